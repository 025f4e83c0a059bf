#!/bin/bash
# Round measurement set, run on the GPU box through gpurun:  gpurun -- 'bash tools/measure.sh r02a [quick]'
# Writes raw output under gpurun_out/<tag>/ ; tools/summarize_profile.py condenses it into profiles/<tag>_*.
TAG=${1:-rXX}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 bench.py                                          > $OUT/bench_atrium1m.json       2> $OUT/bench_atrium1m.err
if [ -z "$QUICK" ]; then
python3 bench.py --workload atrium262k --no-cpu-baseline  > $OUT/bench_c2_atrium262k.json  2>> $OUT/bench_atrium1m.err
python3 bench.py --workload spheres10k                    > $OUT/bench_c3_spheres10k.json  2>> $OUT/bench_atrium1m.err
python3 bench.py --max-path-length 3                      > $OUT/bench_c4_path3.json       2>> $OUT/bench_atrium1m.err
python3 bench.py --identical-frames --no-cpu-baseline     > $OUT/bench_atrium1m_identical_frames.json 2>> $OUT/bench_atrium1m.err
fi
[ -x tools/probes/mem_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/probes/mem_probe tools/probes/mem_probe.hip 2> $OUT/mem_probe_build.err
tools/probes/mem_probe > $OUT/mem_probe.json 2> $OUT/mem_probe.err
# profiler passes: the program itself after `--`, kernel trace and counters in separate runs.  The profiled command is the DEFAULT
# configuration of bench.py (one render() per frame over 8 frame slots, 16 views), shortened, without the oracle leg and the extra modes
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-modes"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
# the same trace with ONE frame at a time: kernels of different frames do not overlap, so the per-kernel average durations of this summary
# are kernel properties and must agree with roofline.contract.per_kernel.*.ms (HIP events) of the bench line written next to it
B1="python3 $R/bench.py --steps 48 --warmup 8 --frames-in-flight 1 --no-cpu-baseline --no-modes"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_one -- $B1 > $OUT/bench_one_at_a_time_under_rocprof.json 2> $OUT/trace_one.err
pmc() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- $B > /dev/null 2> $OUT/pmc_$name.err || echo "pmc pass $name failed" >> $OUT/pmc_failures.txt
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
pmc valu SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pmc tcp2 TCP_TOTAL_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum
pmc tcp3 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pmc tcc TCC_HIT_sum TCC_MISS_sum
pmc tcc2 TCC_REQ_sum TCC_READ_sum
pmc ta TA_BUSY_avr TA_TA_BUSY_sum
pmc ta2 TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
pmc wait SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
rocprofv3 -L > $OUT/counters_available.txt 2>&1
cd $R
python3 tools/summarize_profile.py $TAG $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_inst $OUT/pmc_valu $OUT/pmc_tcp $OUT/pmc_tcp2 $OUT/pmc_tcp3 $OUT/pmc_tcc $OUT/pmc_tcc2 $OUT/pmc_ta $OUT/pmc_ta2 $OUT/pmc_wait
cp $(find $OUT/trace_one -name '*_kernel_stats.csv' | head -1) profiles/${TAG}_kernel_stats_one_at_a_time.csv 2>/dev/null
for f in bench_atrium1m bench_c2_atrium262k bench_c3_spheres10k bench_c4_path3 bench_atrium1m_identical_frames bench_under_rocprof bench_one_at_a_time_under_rocprof mem_probe; do [ -s $OUT/$f.json ] && cp $OUT/$f.json profiles/${TAG}_$f.json; done
# only the small condensed files travel back: drop the raw traces beyond the csv summaries
find $OUT -name '*.db' -delete 2>/dev/null
find $OUT -name '*_counter_collection.csv' -size +20M -delete 2>/dev/null
ls -la profiles/ | tail -14
head -c 400 $OUT/bench_atrium1m.json; echo
