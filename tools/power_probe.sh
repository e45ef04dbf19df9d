# read-only: what the board reports while the hot kernels run (power cap, average socket power, clocks)
mkdir -p gpurun_out
{
rocm-smi --showpower --showmaxpower --showclocks --showperflevel 2>&1 | grep -v "^$" | head -40
echo "== under load (attention backward loop) =="
ITERS=4000 python tools/prof_attn_bwd.py > /dev/null 2>&1 &
python - <<'PY'
import subprocess, time
time.sleep(8)
for i in range(4):
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "-u"], capture_output=True, text=True).stdout
    print("\n".join(l for l in out.splitlines() if any(k in l for k in ("Power", "sclk", "mclk", "GPU use", "busy"))))
    time.sleep(1.0)
PY
wait
} > gpurun_out/power_probe.txt 2>&1
cat gpurun_out/power_probe.txt
