#!/bin/bash
# Round measurement set, run on the GPU box through gpurun:  gpurun -- 'bash tools/measure.sh r03f [quick]'
# For EVERY configuration of BASELINE.json that fits one GPU — c4 (headline: 1 M triangles, primary + shadow), c2 (262 k triangles),
# c3 (10 000 animated instances), c4path (c4 path traced, max path length 3) — it leaves under profiles/:
#   <tag>_<cfg>_bench.json                          the bench line (default mode of bench.py: frames in flight)
#   <tag>_<cfg>_kernel_stats_one_at_a_time.csv      rocprofv3 --kernel-trace --stats with ONE frame at a time: kernels of different frames do
#                                                   not overlap, so these average durations are kernel properties and agree with
#                                                   roofline.contract.per_kernel.*.ms (HIP events) of <tag>_<cfg>_bench_one_at_a_time.json
#   <tag>_<cfg>_kernel_stats_frames_in_flight.csv   the same trace of the DEFAULT command (12 frame slots; C3: 16): durations include the time a
#                                                   kernel shares the chip with the other frames' kernels — not kernel properties
#   <tag>_<cfg>_pmc.json / _pmc_valu.json / _pmc_cache.json   counters per launch (separate --pmc passes), each carrying the hash of the
#                                                   kernel sources they were measured on: bench.py refuses them when the sources changed
# Raw output: gpurun_out/<tag>/ .  `quick` = headline configuration only.  WITH_32M=1 adds c32m (atrium32m, procedural: ~10 more minutes).
TAG=${1:-rXX}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/probes/mem_probe tools/probes/mem_probe.hip 2> $OUT/mem_probe_build.err
tools/probes/mem_probe > $OUT/mem_probe.json 2> $OUT/mem_probe.err
[ -s $OUT/mem_probe.json ] && cp $OUT/mem_probe.json profiles/${TAG}_mem_probe.json

profile_config() { # name, bench arguments...
  local cfg=$1; shift
  local O=$OUT/$cfg
  mkdir -p $O
  cd $R
  python3 bench.py "$@" > $O/bench.json 2> $O/bench.err
  # profiler passes: the program itself after `--`, kernel trace and counters in separate runs, shortened, without the oracle leg and the extra modes
  cd /tmp; export TMPDIR=/tmp
  local B="python3 $R/bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-modes $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/bench_under_rocprof.json 2> $O/trace.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_one -- $B --frames-in-flight 1 > $O/bench_one_at_a_time.json 2> $O/trace_one.err
  pmc() { # name, counters...
    local name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -- $B --frames-in-flight 1 > /dev/null 2> $O/pmc_$name.err || echo "$cfg: pmc pass $name failed" >> $OUT/pmc_failures.txt
  }
  pmc fetch FETCH_SIZE
  pmc write WRITE_SIZE
  pmc inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
  pmc valu SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  pmc tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
  pmc tcc TCC_HIT_sum TCC_MISS_sum
  pmc ta TA_BUSY_avr TA_TA_BUSY_sum
  if [ "$cfg" = "c4" ]; then
    pmc tcp2 TCP_TOTAL_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum
    pmc tcp3 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
    pmc tcc2 TCC_REQ_sum TCC_READ_sum
    pmc ta2 TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
    pmc wait SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
  fi
  cd $R
  python3 tools/summarize_profile.py ${TAG}_$cfg $O/trace_one $O/pmc_fetch $O/pmc_write $O/pmc_inst $O/pmc_valu $O/pmc_tcp $O/pmc_tcc $O/pmc_ta $O/pmc_tcp2 $O/pmc_tcp3 $O/pmc_tcc2 $O/pmc_ta2 $O/pmc_wait
  cp $(find $O/trace -name '*_kernel_stats.csv' | head -1) profiles/${TAG}_${cfg}_kernel_stats_frames_in_flight.csv 2>/dev/null
  for f in bench bench_under_rocprof bench_one_at_a_time; do [ -s $O/$f.json ] && cp $O/$f.json profiles/${TAG}_${cfg}_$f.json; done
  # only the condensed files travel back (gpurun brings at most 64 MiB home, and nothing at all beyond that): the raw traces and counter
  # collections have been summarised above
  rm -rf $O/trace $O/trace_one $O/pmc_*/
}

profile_config c4
if [ -z "$QUICK" ]; then
  profile_config c2 --workload atrium262k
  profile_config c3 --workload spheres10k
  profile_config c4path --max-path-length 3
  # with frames in flight the bounces run the streaming kernels (one frame at a time — which is how counters are taken — they do not): the
  # same passes with streaming forced, for the timed region's ceilings
  RFW_STREAM_RUN=8 profile_config c4pathS --max-path-length 3
  # far outside every cache: the same atrium at 33.5 M triangles (the one configuration where the "% of HBM roofline" clause has a meaning)
  if [ -n "$WITH_32M" ]; then profile_config c32m --workload atrium32m --procedural --no-cpu-baseline; fi
  cd $R
  python3 bench.py --identical-frames --no-cpu-baseline > $OUT/bench_atrium1m_identical_frames.json 2> $OUT/bench_identical.err
  [ -s $OUT/bench_atrium1m_identical_frames.json ] && cp $OUT/bench_atrium1m_identical_frames.json profiles/${TAG}_c4_bench_identical_frames.json
fi
cd $R
rocprofv3 -L > $OUT/counters_available.txt 2>&1
# the counters are in: the bench lines of the round, now WITH the ceilings of their configuration (bench.py reads profiles/<tag>_<cfg>_pmc*.json)
python3 bench.py > $OUT/bench_c4_final.json 2> $OUT/bench_c4_final.err && cp $OUT/bench_c4_final.json profiles/${TAG}_c4_bench.json
if [ -z "$QUICK" ]; then
  python3 bench.py --workload atrium262k > $OUT/bench_c2_final.json 2>> $OUT/bench_c4_final.err && cp $OUT/bench_c2_final.json profiles/${TAG}_c2_bench.json
  python3 bench.py --workload spheres10k > $OUT/bench_c3_final.json 2>> $OUT/bench_c4_final.err && cp $OUT/bench_c3_final.json profiles/${TAG}_c3_bench.json
  python3 bench.py --max-path-length 3 > $OUT/bench_c4path_final.json 2>> $OUT/bench_c4_final.err && cp $OUT/bench_c4path_final.json profiles/${TAG}_c4path_bench.json
  if [ -n "$WITH_32M" ]; then python3 bench.py --workload atrium32m --procedural --no-cpu-baseline > $OUT/bench_c32m_final.json 2>> $OUT/bench_c4_final.err && cp $OUT/bench_c32m_final.json profiles/${TAG}_c32m_bench.json; fi
fi
# the driver's own command (one fill-and-drain of the frame slots) and the builder's timings, for BASELINE.md / DESIGN.md
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_c4_20_steps.json 2> $OUT/bench_c4_20_steps.err && cp $OUT/bench_c4_20_steps.json profiles/${TAG}_c4_bench_20_steps.json
if [ -z "$QUICK" ]; then
  python3 tools/probes/build_time.py 0 > $OUT/build_time.json 2> $OUT/build_time.err && cp $OUT/build_time.json profiles/${TAG}_build_time.json
fi
# kernel timelines (start / end of every kernel of the timed region): the driver's command, and C3 with its per-frame instance update
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_c4 -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-modes > $OUT/tl_c4_line.json 2> $OUT/tl_c4.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_c3 -- python3 $R/bench.py --gpus 1 --workload spheres10k --steps 60 --warmup 24 --no-cpu-baseline --no-modes > $OUT/tl_c3_line.json 2> $OUT/tl_c3.err
cd $R
python3 tools/timeline.py $OUT/tl_c4 20 5 profiles/${TAG}_c4_20_steps_timeline.csv > profiles/${TAG}_c4_20_steps_timeline.json 2>> $OUT/tl_c4.err
python3 tools/timeline.py $OUT/tl_c3 60 24 profiles/${TAG}_c3_timeline.csv > profiles/${TAG}_c3_timeline.json 2>> $OUT/tl_c3.err
rm -rf $OUT/tl_c4 $OUT/tl_c3
# gpurun only brings gpurun_out/ back: the condensed files ride along in it (copy them into profiles/ of the checkout afterwards)
mkdir -p $OUT/profiles && cp profiles/${TAG}_* $OUT/profiles/
ls -la profiles/ | tail -40
head -c 400 $OUT/bench_c4_final.json; echo
