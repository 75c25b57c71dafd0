"""Where a level of the wide-band sweep spends its time: run one pair on a BIALIGN_EXP=8 build (tools/exp_build.sh), whose
part 0 prints s_memtime ticks per phase (100 MHz on gfx950: 1 tick = 10 ns).  WIDE_LEN / WIDE_S."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch
n, s = int(os.environ.get("WIDE_LEN", 1000)), int(os.environ.get("WIDE_S", 6))
for so in (False, True):
    b = make_batch([synth.rna_pair(2, n, n)], dict(synth.RNA_PARAMS, max_shift=s), score_only=so)
    b.run(); t = b.timing()
    print(f"{n}x{n} s={s} {'score-only' if so else 'full'}: fill {t['fill_ms']:.2f} ms, waves/pair {t['waves_per_pair']}", flush=True)
    b.close()
