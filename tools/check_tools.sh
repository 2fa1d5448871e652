#!/bin/bash
# Runs every probe under tools/ (and tests/tools/) once, small, against the library at HEAD; one line per tool: ok / FAILED.
# On the GPU box, through gpurun; the variant builds k2_trace, er_trace and em_trace must exist (tools/variants.sh k2_trace "MC_K2_TRACE=1"
# er_trace "MC_ER_TRACE=1" em_trace "MC_EM_TRACE=1" fd_trace "MC_FD_TRACE=1" fd_stop4 "MC_FD_STOP=4").  The log of the round's run:
# profiles/r06_tools_check.log.
export TMPDIR=/tmp
out=gpurun_out/tools_check; mkdir -p $out
run() { name=$1; shift; if timeout 300 "$@" > $out/$name.log 2>&1; then echo "ok      $name"; else echo "FAILED  $name (exit $?): $(tail -n 1 $out/$name.log | cut -c1-120)"; fi; }
run file_to_file        python3 tools/file_to_file.py 2e6 --runs 2
run file_to_file_gpus2  env MCALLER_SHARD_DEVICES=0,0 python3 tools/file_to_file.py 2e6 --runs 2 --gpus 2 --bed
run f2f_cprofile        python3 tools/f2f_cprofile.py 2e6
run f2f_profile         tools/f2f_profile.sh 2e6
run kstats              tools/kstats.sh tools_check/kstats --events 2e7 --steps 6
run sparse_stats        tools/sparse_stats.sh tools_check/sparse --events 2e7 --steps 6
run dense_stats         tools/dense_stats.sh tools_check/dense --events 2e7
run timeline            tools/timeline.sh
run pipeline_probe      python3 tools/pipeline_probe.py 2e7
run pipeline_soak       python3 tools/pipeline_soak.py 2e7 40
run stream_soak         python3 tools/stream_soak.py 1e6 3
run shard_probe         python3 tools/shard_probe.py
run parse_threads_probe python3 tools/parse_threads_probe.py
run pread_probe         python3 tools/pread_probe.py
run read_probe          python3 tools/read_probe.py
run dma_probe           python3 tools/dma_probe.py
run forest_probe        python3 tools/forest_probe.py 2e7
run variant_probe       python3 tools/variant_probe.py
run k2_trace            env MCALLER_LIB=mcaller_amd/variants/k2_trace.so python3 tools/k2_trace.py 2e7
run side_trace          env MCALLER_LIB=mcaller_amd/variants/k2_trace.so python3 tools/side_trace.py 2e7
run side_trace_dense    env MCALLER_LIB=mcaller_amd/variants/k2_trace.so python3 tools/side_trace.py 2e7 A
run kres                python3 tools/kres.py mc_fused
run rowtext_probe       python3 tools/rowtext_probe.py 3e5 3
run fuzz_rowtext        python3 tests/tools/fuzz_rowtext.py 6
run er_trace            env MCALLER_LIB=mcaller_amd/variants/er_trace.so python3 tools/er_trace.py 1e8     # (its trace window is workgroups 40000-41023: 10^8 rows)
run em_trace            env MCALLER_LIB=mcaller_amd/variants/em_trace.so python3 tools/em_trace.py 2e7
run emit_grid_probe     tools/emit_grid_probe.sh 2e7
run fd_trace            env MCALLER_LIB=mcaller_amd/variants/fd_trace.so python3 tools/fd_trace.py 1e8     # (its trace window is workgroups 40000-41023: 10^8 rows)
run fused_probe         python3 tools/fused_probe.py 2e7
run file_to_file_dense  python3 tools/file_to_file.py 2e6 --runs 2 --motif A
run config5             python3 tools/config5.py 2000000 --runs 2
run project_scaling     python3 tools/project_scaling.py --rows 2e6 --parts 2 --runs 2
run side_probe          python3 tools/side_probe.py 2e7
run fused_phase_counters tools/fused_phase_counters.sh 2e7
run fuzz_gpu            python3 tests/tools/fuzz_gpu.py 5 71000000
run fuzz_tables_dense   python3 tests/tools/fuzz_tables.py 20 72000000 dense
run fuzz_tables_sparse  python3 tests/tools/fuzz_tables.py 20 72000000 sparse
run train_probe         python3 tests/tools/train_probe.py
for m in tools/micro/*.hip; do b=$(basename $m .hip); if /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mc_micro_$b $m > $out/micro_$b.log 2>&1 && timeout 120 /tmp/mc_micro_$b >> $out/micro_$b.log 2>&1; then echo "ok      micro/$b"; else echo "FAILED  micro/$b: $(tail -n 1 $out/micro_$b.log | cut -c1-120)"; fi; done
