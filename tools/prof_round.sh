# usage (on the GPU box, from the repo root): bash tools/prof_round.sh [round]
# collect the round's evidence on the GPU box: bench line, kernel-trace stats (default = weight gradients on a side stream, and
# serial = one kernel at a time, the mode bench.py's roofline pass uses), PMC passes in the serial mode (separate, as the guide prescribes)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=${1:-6}
O=gpurun_out/r0${R}p
mkdir -p $O
python3 bench.py --steps 10 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
export UC2_WGRAD_SIDE=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/trace_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_mlm -- python3 bench.py --steps 6 --warmup 8 --no-cpu-baseline --no-extras --task mlm > $O/trace_mlm.log 2>&1
B="python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
python3 tools/pmc_summarize.py $R $O/pmc_fetch $O/pmc_write $O/pmc_sq "gemm_bf16_pp16_kernel<true, true, true, 0, 2>" "gemm_bf16_pp16_kernel<false, false, true, 5, 2>" "gemm_bf16_pp16_kernel<false, false, true, 6, 2>" "gemm_bf16_pp16_kernel<false, false, true, 3, 2>" "gemm_bf16_pp16_kernel<false, false, true, 0, 2>" "gemm_bf16_pp16_kernel<false, false, true, 10, 2>" attn_bwd_mfma attn_fwd_mfma ln_bwd_kernel ln_fwd16_kernel adamw_kernel > $O/pmc_summary.json
unset UC2_WGRAD_SIDE
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_1024 -- python3 bench.py --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/trace_1024.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_2048 -- python3 bench.py --batch 2048 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/trace_2048.log 2>&1
python3 scratch/regime_step.py itm 8 > $O/regime_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_regime -- python3 scratch/regime_step.py itm 8 > $O/trace_regime.log 2>&1
python3 tools/timeline.py $O/trace_regime adamw_kernel 4 > $O/regime_timeline.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_large_bf16 -- python3 scratch/large_step.py bf16 > $O/trace_large_bf16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_large_fp8 -- python3 scratch/large_step.py fp8 > $O/trace_large_fp8.log 2>&1
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*.db" -delete
du -sh $O
