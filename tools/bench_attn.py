#!/usr/bin/env python3
"""Decoder attention kernels at the S4 shapes (BT = 40 frames x 8 heads, 100 queries): forward / backward times after a
~0.3 s warm-up (sustained clocks), per key length.  `rocprofv3 --kernel-trace --stats -- python3 tools/bench_attn.py` gives the
per-kernel durations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import combo_avs_amd  # noqa: F401
from combo_avs_amd.ops.attention import attention

B, H, E, Lq = 40, 8, 256, 100
torch.manual_seed(0)


def t(fn, secs=0.3, n=50):
    t0 = time.time()
    while time.time() - t0 < secs:
        fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


from combo_avs_amd import _lib
import ctypes
lib = _lib.lib()


def device_us(fn, kinds=(4,), n=30):
    """mean device-side duration (kernel timing slots, csrc/timing.hip) of the instrumented kernels of `kinds` in fn()"""
    buf = torch.zeros(4096, 256, dtype=torch.int64, device="cuda")
    buf[:, 0::16] = -1
    torch.cuda.synchronize()
    lib.combo_timing_set_buffer(ctypes.c_void_p(buf.data_ptr()), 4096)
    for _ in range(n):
        fn()
    lib.combo_timing_fold(_lib.current_stream())  # (every launch of this loop has its own slot)
    torch.cuda.synchronize()
    used = lib.combo_timing_slots_used()
    khz = float(lib.combo_wall_clock_khz())
    lib.combo_timing_slot_info.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]
    tsv = buf.cpu()
    out = {}
    for sl in range(used):
        kind, work = ctypes.c_int(0), ctypes.c_double(0)
        lib.combo_timing_slot_info(sl, ctypes.byref(kind), ctypes.byref(work))
        if int(tsv[sl, 3]) > 0:
            out.setdefault(kind.value, []).append(float(tsv[sl, 2]) / khz * 1e3 / int(tsv[sl, 3]))
    lib.combo_timing_set_buffer(None, 0)
    return out


for Lk in [int(x) for x in (sys.argv[1:] or ["784", "196", "49", "100"])]:
    q = torch.randn(B * Lq, E, device="cuda", requires_grad=True)
    k = torch.randn(B * Lk, E, device="cuda", requires_grad=True)
    v = torch.randn(B * Lk, E, device="cuda", requires_grad=True)
    pitch = (Lk + 3) // 4 * 4
    blocked = (torch.rand(B, Lq, pitch, device="cuda") < 0.5).to(torch.uint8)
    blocked[:, :, 0] = 0
    if Lk == 100:
        blocked = None
    fwd = lambda: attention(q, k, v, blocked, B, H)
    out = fwd()
    g = torch.randn_like(out)
    with torch.no_grad():
        tf = t(fwd)
    tfb = t(lambda: torch.autograd.grad(fwd(), (q, k, v), g))
    # back-to-back launches through the C ABI (no autograd / allocator in the loop): stream-serialised, so the mean is the
    # kernel duration + the inter-kernel gap
    out_b = torch.empty(B * Lq, E, device="cuda"); lse_b = torch.empty(B, H, Lq, device="cuda")
    st = _lib.current_stream()
    raw = lambda: lib.combo_attention_forward_f32(q.data_ptr(), E, k.data_ptr(), E, v.data_ptr(), E, _lib.ptr(blocked), pitch if blocked is not None else 0, None, 0,
                                                  B, H, Lq, Lk, 32 ** -0.5, out_b.data_ptr(), lse_b.data_ptr(), st)
    print(f"[attention Lk={Lk}] forward, back-to-back C-ABI launches: {t(raw, n=200):.1f} us", flush=True)
    junk = torch.empty(64 << 20, device="cuda")  # 256 MB: evicts L2 / MALL / instruction caches between launches
    def raw_cold():
        junk.add_(1.0)
        raw()
    d1 = device_us(raw, n=50).get(4, [0])
    d2 = device_us(raw_cold, n=50).get(4, [0])
    print(f"[attention Lk={Lk}] forward device-side: back-to-back {sum(d1) / len(d1):.1f} us, after a 256 MB sweep {sum(d2) / len(d2):.1f} us", flush=True)
    dv = device_us(lambda: torch.autograd.grad(fwd(), (q, k, v), g))
    mean = lambda x: sum(x) / max(len(x), 1)
    print(f"[attention Lk={Lk}] device-side: forward {mean(dv.get(4, [])):.1f} us, backward dq {mean(dv.get(5, [])[0::2]):.1f} us, "
          f"dk/dv {mean(dv.get(5, [])[1::2]):.1f} us", flush=True)
    print(f"[attention Lk={Lk}] forward {tf:.1f} us ({4.0 * B * H * Lq * Lk * 32 / tf / 1e6:.1f} TFLOP/s), forward+backward {tfb:.1f} us", flush=True)
