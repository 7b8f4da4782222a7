"""Times the discriminant launch of one scheme alone (tams, ell=3, 10^4 frequencies): HIP-event kernel time per launch.
usage: time_det.py scheme [reps] [n_freq]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storm_amd import host
scheme = sys.argv[1] if len(sys.argv) > 1 else "magnus6"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_tams.npz"))
s = host.ErasedSolver(model, 3, 0, scheme, model.field("r_coord"))
s.set_option("profile", 1)
f = host.linspace(1.0, 25.0, n)
d = s.det(f)
s.kernel_times(reset=True)
for _ in range(reps):
    d = s.det(f)
kt = {k: v for k, v in s.kernel_times(reset=True).items() if v[1]}
import hashlib
print(scheme, n, {k: round(ms / cnt, 4) for k, (ms, cnt) in kt.items()}, hashlib.md5(d.tobytes()).hexdigest())
