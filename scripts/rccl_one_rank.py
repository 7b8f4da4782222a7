"""One rank, backend nccl (= RCCL): the strong-scaling exchange path of storm_amd/distributed.py on device tensors.
With one rank the gathers are trivial, but the communicator, the single-tensor all_gather form and the pinned mirrors are the
ones N ranks would use.  Prints the phases and checks the result against the one-call scan."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
from storm_amd import distributed as D, ffi, host

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print(f"init_process_group(nccl) {time.perf_counter() - t0:.2f} s", flush=True)
try:
    model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_zams.npz"))
    es = host.ErasedSolver(model, 2, 0, "colloc4", model.field("r_coord"))
    h = ffi.HipSolver.borrow(es.raw_handle(), owner=es)
    f = host.linspace(1.0, 25.0, 4000)
    ulp = ffi.relative_to_ulp(1e-8)
    ref, _, _ = h.scan_and_optimize(f, ulp, cap=4096, want_dets=False)
    for k in range(3):
        t = {}
        t0 = time.perf_counter()
        res, inv, mine = D.strong_scan_and_optimize(h, f, ulp, device=dev, timings=t, records=True, deal=False)
        dt = time.perf_counter() - t0
        print(f"pass {k}: {dt * 1e3:.3f} ms", {k2: round(v * 1e3, 3) for k2, v in t.items()}, flush=True)
    ref_rec = D.pack(ref, 0)
    assert res.shape == ref_rec.shape and res.tobytes() == ref_rec.tobytes(), "records differ from the one-call scan"
    # the exchange itself (a one-rank scan skips it): records -> pinned mirror -> device tensor -> RCCL all_gather_into_tensor ->
    # device tensor -> pinned mirror, in the one-collective form and in the two-collective form of an oversize count
    got, any_inv = D.gather_brackets(ref_rec, False, dev)
    assert got.tobytes() == ref_rec.tobytes() and any_inv is False
    big = np.arange((D.GATHER_CAP + 37) * D.RECORD, dtype=np.float64).reshape(-1, D.RECORD)
    got, any_inv = D.gather_brackets(big, True, dev)
    assert got.tobytes() == big.tobytes() and any_inv is True
    for k in range(3):
        t0 = time.perf_counter()
        D.gather_brackets(ref_rec, False, dev)
        print(f"gather_brackets over RCCL, one rank: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
    assert D._single_tensor_form() is True
    x = torch.ones(4, device=dev, dtype=torch.float64)
    dist.all_reduce(x)
    dist.barrier()
    torch.cuda.synchronize()
    print("ok", len(ref), "brackets; form:", D._single_tensor_form() if hasattr(D, "_single_tensor_form") else "?")
finally:
    dist.destroy_process_group()
