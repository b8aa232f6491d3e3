# rocprofv3 kernel trace of generate() (4 images) for the e4m3 and the 16-bit model, summarised for the LAST decode step:  bash tests/diag/r06_decode_step.sh <tag>
T=${1:-r06h}; O=gpurun_out/$T; mkdir -p $O; R=$GRAFT_REPO_ROOT
for v in fp8 fp16; do
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -o gen -- python $R/bench.py --dtype $v --mode generate --batch 4 --steps 2 --warmup 2 --no-cpu-baseline --no-traffic --no-parity > $R/$O/bench_gen_prof_$v.json 2>/dev/null
  cd $R
  G=$(find $O/prof_$v -name "*kernel_trace.csv" | head -1)
  python tests/diag/decode_trace.py $G > $O/decode_step_$v.txt 2>&1
  rm -rf $O/prof_$v
  cat $O/decode_step_$v.txt
done
