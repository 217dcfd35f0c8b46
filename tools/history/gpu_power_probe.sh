# socket power and clocks (rocm-smi) while (a) the HBM-sized UR5 OSC kernel, (b) the same kernel on a cache-resident batch,
# (c) the register-only fp64 FMA microbenchmark run back to back for a few seconds each -> gpurun_out/power_probe.txt
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/power_probe.txt
hipcc --offload-arch=gfx950 -O3 tools/microbench/sustained_rates.hip -o /tmp/sustained 2>/dev/null
cat > /tmp/loop.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
import abr_control_amd as a
from abr_control_amd import _abi, engine
from abr_control_amd._lib import check, lib
B, secs, useC = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
arm = check(lib().abrk_arm_builtin(b"ur5"))
rng = np.random.RandomState(1)
q, dq, t = rng.uniform(0, 2*np.pi, (B, 6)), rng.uniform(0, 5, (B, 6)), rng.uniform(-1, 1, (B, 6))
s = a.Stream(0)
qd, dd, td = (a.DeviceArray.from_numpy(x) for x in (q, dq, t)); u = a.DeviceArray((B, 6))
with engine.Plan(0, s) as plan:
    engine.osc_generate(arm, 6, _abi.make_osc_params(6, kp=200, use_C=bool(useC)), qd, dd, td, u=u, stream=s)
n = max(1, (1 << 23) // B)
t0 = time.time(); k = 0
while time.time() - t0 < secs:
    tt = time.time(); plan.launch_graph(10 * n); s.sync(); k += 1; last = (time.time() - tt) / (10 * n)
print(f"B={B} use_C={useC}: last {last*1e6:.1f} us per launch = {B/last/1e9:.2f} G rows/s", flush=True)
PY
sample() { for i in 1 2 3 4; do sleep 0.7; rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "power|sclk" | sed 's/^/    /'; echo "    --"; done; }
{
echo "== idle"; rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "power|sclk"
echo "== UR5 OSC xyz+g, 8M rows (HBM-sized), back to back"; python /tmp/loop.py 8388608 4 0 & sleep 1.2; sample; wait
echo "== UR5 OSC xyz+g, 262144 rows (50 MB: cache-resident), back to back"; python /tmp/loop.py 262144 4 0 & sleep 1.2; sample; wait
echo "== UR5 OSC xyz+g+C, 8M rows"; python /tmp/loop.py 8388608 4 1 & sleep 1.2; sample; wait
echo "== register-only fp64 FMA chains (tools/microbench/sustained_rates.hip)"; /tmp/sustained > /tmp/sus.txt & sleep 0.5; sample; wait; head -2 /tmp/sus.txt
} > $O 2>&1
cat $O | head -120
