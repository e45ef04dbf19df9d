# same-box A/B of this tree against the round-4 tree (ab/r04tree = git worktree of 836d18a with its library built)
R=$PWD; mkdir -p gpurun_out
{
for rep in 1 2; do
for cfg in "" "--config 3" "--config 4"; do
  for tree in . ab/r04tree; do
    v=$(cd $R/$tree && timeout 600 python bench.py $cfg --no-cpu-baseline --exact-f32 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "rep $rep  bench.py $cfg  tree $tree : $v"
  done
done
done
} > gpurun_out/ab_r04.log 2>&1
cat gpurun_out/ab_r04.log
