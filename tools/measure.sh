#!/bin/bash
# Round measurement set, run on the GPU box through gpurun:  gpurun -- 'bash tools/measure.sh r01d'
# Writes raw output under gpurun_out/<tag>/ ; tools/summarize_profile.py condenses it into profiles/<tag>_*.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 bench.py                                          > $OUT/bench_atrium1m.json       2> $OUT/bench_atrium1m.err
python3 bench.py --frames-in-flight 1 --batch 1 --no-cpu-baseline   > $OUT/bench_atrium1m_f1.json    2>> $OUT/bench_atrium1m.err
python3 bench.py --batch 1 --frames-in-flight 8 --no-cpu-baseline  > $OUT/bench_atrium1m_render_per_frame.json 2>> $OUT/bench_atrium1m.err
python3 bench.py --workload atrium262k --no-cpu-baseline  > $OUT/bench_c2_atrium262k.json  2>> $OUT/bench_atrium1m.err
python3 bench.py --workload spheres10k --no-cpu-baseline  > $OUT/bench_c3_spheres10k.json  2>> $OUT/bench_atrium1m.err
python3 bench.py --max-path-length 3 --no-cpu-baseline    > $OUT/bench_c4_path3.json       2>> $OUT/bench_atrium1m.err
# profiler passes: the program itself after `--`, kernel trace and counters in separate runs
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --frames-in-flight 1 --batch 1 --steps 20 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_inst -- $B > /dev/null 2> $OUT/pmc_inst.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_valu -- $B > /dev/null 2> $OUT/pmc_valu.err
cd $R
python3 tools/summarize_profile.py $TAG $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_inst $OUT/pmc_valu
for f in bench_atrium1m bench_atrium1m_f1 bench_atrium1m_render_per_frame bench_c2_atrium262k bench_c3_spheres10k bench_c4_path3 bench_under_rocprof; do cp $OUT/$f.json profiles/${TAG}_$f.json; done
# only the small condensed files travel back: drop the raw traces beyond the csv summaries
find $OUT -name '*.db' -delete 2>/dev/null
ls -la profiles/ | tail -12
head -c 400 $OUT/bench_atrium1m.json; echo
