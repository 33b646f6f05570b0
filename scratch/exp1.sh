mkdir -p gpurun_out/r02b
python tests/bench_pp_epi.py > gpurun_out/r02b/pp_epi.log 2>&1
for sk in "" "1:1,2:1" "1:2,2:2" "1:3,2:3" "0:1,1:1,2:1,3:1"; do
  UC2_PP_SKEW=$sk python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r02b/step_skew_$sk.json 2>/dev/null
done
