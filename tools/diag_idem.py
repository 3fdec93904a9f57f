"""Diagnostic: idempotence failures of the fp32 path (not collected by pytest)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from sccav_cbf_amd import scenarios
from sccav_cbf_amd.batched import BatchedCBF
np.set_printoptions(precision=9, linewidth=160)
dev = torch.device("cuda:0")
w = scenarios.config4(A=65536, nominal="uniform")
t = {k: torch.from_numpy(v).to(dev) for k, v in w.items()}
for k in ("state", "u_ref", "obs"):
    t[k] = t[k].float().contiguous()
f = BatchedCBF(alpha=1.0, R=np.diag([0.5, 0.5]), slip_io=True)
u, st, rows = f.solve(t["state"], t["u_ref"], t["obs"], t["obs_type"], want_rows=True)
u2, st2 = f.solve(t["state"], u.contiguous(), t["obs"], t["obs_type"])
torch.cuda.synchronize()
ok = (st == 0)
bad = torch.nonzero(ok & (st2 != 0)).flatten().cpu().numpy()
print("failing", len(bad), bad[:20])
p = orc.make_params(alpha=1.0, R=(0.5, 0, 0, 0.5), flags=orc.FLAG_SLIP_IO)
rows_c = rows.double().cpu().numpy(); u_c = u.double().cpu().numpy()
for a in bad[:6]:
    r = rows_c[:, a, :]
    print("agent", a, "st2", int(st2[a]), "u", u_c[a], "uref", w["u_ref"][a])
    l = np.sqrt(0.5)
    at = r[:, :2] / l; n = np.hypot(at[:, 0], at[:, 1]) + 1e-300
    bt = (r[:, 2] - r[:, :2] @ u_c[a]) / n
    print("  unit normals", np.round((at / n[:, None]), 6).tolist())
    print("  bt at u    ", bt.tolist())
    uo, so = orc.qp_batch(orc.make_params(R=(0.5, 0, 0, 0.5)), np.ascontiguousarray(r[:, None, :]), u_c[a][None, :])
    print("  oracle on fp32 rows from u:", so, uo)
