"""one allocation + views against nine torch.empty calls for the module path's output tensors (VERDICT r4 #3b): host time per call"""
import time, torch
dev = torch.device("cuda:0")
B, N0, F0, Z, C = 64, 4998, 3, 16, 2
f32 = {"dtype": torch.float32, "device": dev}
def nine():
    return (torch.empty((), dtype=torch.float64, device=dev), torch.empty((), dtype=torch.int64, device=dev),
            torch.empty(B, N0, F0, **f32), torch.empty(B, **f32), torch.empty(B, dtype=torch.float64, device=dev),
            torch.empty(B, Z, **f32), torch.empty(B, C, **f32), torch.empty(B, Z, **f32), torch.empty(B, Z, **f32))
sizes = [B * N0 * F0, B, B * Z, B * C, B * Z, B * Z]
offs, o = [], 0
for s in sizes:
    offs.append(o); o += -(-s // 64) * 64
tot = o + 64 + 2 * B + 64
shapes = [((B, N0, F0), (N0 * F0, F0, 1)), ((B,), (1,)), ((B, Z), (Z, 1)), ((B, C), (C, 1)), ((B, Z), (Z, 1)), ((B, Z), (Z, 1))]
def one():
    buf = torch.empty(tot, **f32)
    v = [torch.as_strided(buf, sh, st, of) for (sh, st), of in zip(shapes, offs)]
    d = buf[o:o + 2 * B + 16].view(torch.float64)
    return d[0], buf[o + 2 * B + 32:o + 2 * B + 34].view(torch.int64)[0], v[0], v[1], d[1:1 + B], v[2], v[3], v[4], v[5]
for f in (nine, one, nine, one):
    for _ in range(2000): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20000): f()
    print(f.__name__, f"{(time.perf_counter() - t0) / 20000 * 1e6:.1f} us per call")
