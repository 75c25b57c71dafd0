import sys, json, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bialign_amd import synth
from bialign_amd.batch import make_batch
from bialign_amd.engine import trace_codes_to_columns
from oracle import oracle
recs = json.load(open('' + os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/tests/golden/known_answers.json'))
for rec in recs:
    p = rec['params']
    if p['gap_opening_cost'] == 0 or p['max_shift'] > 3: continue
    b = make_batch([(rec['seqA'], rec['seqB'], rec['strA'], rec['strB'])], p)
    b.run()
    sc = int(b.scores()[0]); tr, ok = b.traces()
    print(rec['name'], 'score', sc, rec['score'], 'trace_ok', trace_codes_to_columns(tr[0]) == rec['trace'], ok[0], rec['complete'], b.timing())
    if 'layers' in rec:
        n,m,s = len(rec['seqA']), len(rec['seqB']), p['max_shift']
        vals = oracle.band_values(b.dump_layers(0), n, m, s)
        bad = sum(int((np.array(g) != np.array(e)).sum()) for g,e in zip(vals, rec['layers']))
        print('   layer mismatches', bad, 'of', sum(len(e) for e in rec['layers']))
