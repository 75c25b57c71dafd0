#!/bin/bash
# counters of the config-2 launch (1024 protein pairs x len 512, s=1) on the final tree, for profiles/hbm_traffic.json
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r03x_cfg2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp AB_LEN=512 AB_PAIRS=1024 AB_STEPS=4
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o stats -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > "$out/stats.log" 2> "$out/stats.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out" -o write -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > /dev/null 2> "$out/write.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out" -o fetch -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > /dev/null 2> "$out/fetch.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$out" -o sq1 -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > /dev/null 2> "$out/sq1.err"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out" -o gui -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > /dev/null 2> "$out/gui.err"
cd $GRAFT_REPO_ROOT && python tools/summarize_profile.py gpurun_out/r03x_cfg2 gpurun_out/r03x_cfg2/summary | grep fill_affine | cut -c1-40,120-260; cat $out/stats.log
