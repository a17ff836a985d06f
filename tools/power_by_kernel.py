"""Dev tool: socket power and shader clock (rocm-smi) while ONE kernel class runs back to back for a few seconds - which classes
pull the bench's power-limited clock down.  Usage (GPU box): python tools/power_by_kernel.py"""
import ctypes as C, os, subprocess, sys, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gyre_amd import _lib
L = _lib.lib(); DEV = "cuda:0"
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
rnd = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)

def smi():
    o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o); p = re.search(r"Power \(W\): ([\d.]+)", o)
    return (int(c.group(1)) if c else 0, float(p.group(1)) if p else 0.0)

def run(name, fn, flops, secs=5.0):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    samples, stop = [], False
    def sampler():
        while not stop:
            samples.append(smi()); time.sleep(0.3)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(50): fn()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop = True; th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    s = samples[len(samples) // 3:]            # steady part
    clk = sum(a for a, _ in s) / max(1, len(s)); pw = sum(b for _, b in s) / max(1, len(s))
    print(f"{name:44s} {us:8.1f} us  {flops / us / 1e6 if flops else 0:6.0f} TFLOP/s  sclk {clk:6.0f} MHz  {pw:6.0f} W", flush=True)
    time.sleep(3)

B = 16
x = rnd(B, 64, 64, 320); w = rnd(320, 9 * 320); b = torch.zeros(320, device=DEV); y = torch.empty(B, 64, 64, 320, dtype=torch.bfloat16, device=DEV)
run("conv 64x64 320->320 (pipelined 256x320)", lambda: L.gyre_op_conv3x3(st(), vp(x), B, 64, 64, 320, vp(w), 320, vp(b), None, 1, 0, 0, vp(y)), 2.0 * B * 4096 * 320 * 2880)
x2 = rnd(B, 64, 64, 960); w2 = rnd(320, 9 * 960)
run("conv 64x64 960->320", lambda: L.gyre_op_conv3x3(st(), vp(x2), B, 64, 64, 960, vp(w2), 320, vp(b), None, 1, 0, 0, vp(y)), 2.0 * B * 4096 * 320 * 8640)
h, D, N = 8, 40, 4096
q = rnd(B, N, h * D); k = (torch.randn(B, N, h * D, device=DEV) * 0.228).to(torch.bfloat16); vt = rnd(B, h * D, N); o = torch.empty(B, N, h * D, dtype=torch.bfloat16, device=DEV)
run("attention 16x8x4096^2 D=40", lambda: L.gyre_op_attention_ex(st(), vp(q), h * D, vp(k), h * D, vp(vt), N, B, h, N, N, D, vp(o), h * D, 1), 4.0 * B * h * N * N * D)
M = 65536
xa = rnd(M, 320); wf = rnd(2560, 320); bf = torch.zeros(2560, device=DEV); yf = torch.empty(M, 1280, dtype=torch.bfloat16, device=DEV)
run("GEGLU FF1 65536x320 -> 2x1280", lambda: L.gyre_op_linear(st(), vp(xa), M, 320, vp(wf), 1280, vp(bf), None, 1, vp(yf)), 2.0 * M * 2560 * 320)
wq = rnd(320, 320); yq = torch.empty(M, 320, dtype=torch.bfloat16, device=DEV); r = rnd(M, 320)
run("projection 65536x320x320 + residual", lambda: L.gyre_op_linear(st(), vp(xa), M, 320, vp(wq), 320, vp(b), vp(r), 0, vp(yq)), 2.0 * M * 320 * 320)
M2 = 16384
xb = rnd(M2, 640); wb = rnd(640, 640); b6 = torch.zeros(640, device=DEV); yb = torch.empty(M2, 640, dtype=torch.bfloat16, device=DEV)
run("projection 16384x640x640 (128x160 tile)", lambda: L.gyre_op_linear(st(), vp(xb), M2, 640, vp(wb), 640, vp(b6), None, 0, vp(yb)), 2.0 * M2 * 640 * 640)
g = torch.ones(320, device=DEV); be = torch.zeros(320, device=DEV)
ws = torch.empty(1 << 24, dtype=torch.uint8, device=DEV)
run("GroupNorm + SiLU 16x64x64x320", lambda: L.gyre_op_groupnorm(st(), vp(x), None, 320, B, 4096, 320, 32, vp(g), vp(be), C.c_float(1e-5), 1, vp(ws), ws.numel(), vp(y)), 0)
# sustained (power-limited) timings of the pipelined conv kernel under its ablations: needs a GYRE_GEMM_ABLATIONS=1 build
if os.environ.get("ABL"):
    for (Ci, xx, ww) in ((320, x, w), (960, x2, w2)):
        for bits, what in ((0, "full"), (0x80, "A requests on 1/9 taps"), (0x200, "A requests on 3/9 taps"), (8, "A from the zero page"), (16, "W from the zero page"),
                           (2, "no operand DMA"), (4, "no epilogue")):
            L.gyre_debug_gemm_ablation(bits)
            run(f"conv {Ci}->320 [{what}]", lambda: L.gyre_op_conv3x3(st(), vp(xx), B, 64, 64, Ci, vp(ww), 320, vp(b), None, 1, 0, 0, vp(y)), 2.0 * B * 4096 * 320 * 9 * Ci, secs=4.0)
    L.gyre_debug_gemm_ablation(0)
