import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import combo_avs_amd
from combo_avs_amd import _lib
from test_sra_gpu import reference
B, N, h, Nk = 2, 3136, 1, 49
torch.manual_seed(B * 1000 + N + h + Nk)
C, scale = 64 * h, 64 ** -0.5
q = (torch.randn(B, N, C, device="cuda") * 1.5).to(torch.bfloat16)
kv = (torch.randn(B, Nk, 2 * C, device="cuda") * 1.5).to(torch.bfloat16)
ref = reference(q, kv, h, scale)
bad_total = 0
for rep in range(20):
    out = torch.full_like(q, float("nan"))
    lse2 = torch.full((B, h, (N + 31) // 32 * 32), float("nan"), device="cuda")
    _lib.check(_lib.lib().combo_sra_attention_forward_bf16(q.data_ptr(), kv.data_ptr(), out.data_ptr(), lse2.data_ptr(), B, N, Nk, h, scale, _lib.current_stream()), "fwd")
    torch.cuda.synchronize()
    err = (out.float() - ref).abs()
    e_t = err.amax(-1).view(B, -1, 32).amax(-1)
    bad = (e_t > 0.05) | torch.isnan(e_t)
    bad_total += int(bad.sum())
    if rep < 3 and bad.any():
        b0, t0 = [int(v) for v in bad.nonzero()[0]]
        rows = (err[b0, t0 * 32:(t0 + 1) * 32] > 0.05)
        print("  example tile", (b0, t0), "bad rows:", rows.any(-1).nonzero().flatten().tolist(), "bad cols of first bad row:", rows[rows.any(-1).nonzero()[0, 0]].nonzero().flatten().tolist())
print("COMBO_SRA_DBG=%s: bad tiles over 20 runs: %d" % (os.environ.get("COMBO_SRA_DBG", "0"), bad_total))
