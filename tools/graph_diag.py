#!/usr/bin/env python3
"""ONE-SHOT diagnosis of the two hipGraph instantiation failures recorded in round 1 (engine.py / vae_step.hip
comments): "a memset node on a forked stream" and "a fork of a fork".  Each scenario is captured through the raw
HIP API (ctypes on the runtime torch has loaded), the captured topology is dumped with hipGraphDebugDotPrint BEFORE
hipGraphInstantiate is called, and the return code of every API call is printed; Python's faulthandler gives the
native backtrace if the runtime dies instead of returning.  Run each scenario in its own process, once:

    python -X faulthandler tools/graph_diag.py <scenario> <out_dir>

scenarios: plain | memset_fork | nested_fork | nested_fork_memset | step_nested (the real two-chain train step)
"""
import ctypes
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mesh-vae_amd"))
import torch  # noqa: E402

scenario, out_dir = sys.argv[1], sys.argv[2]
os.makedirs(out_dir, exist_ok=True)
dev = torch.device("cuda:0")
torch.cuda.init()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = ctypes.c_char_p


def call(name, *args):
    rc = getattr(hip, name)(*args)
    print(f"  {name} -> {rc} ({hip.hipGetErrorString(rc).decode()})", flush=True)
    return rc


a = torch.zeros(1 << 16, device=dev)
b = torch.zeros(1 << 16, device=dev)
c = torch.zeros(1 << 16, device=dev)
main, s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
step_fn = None
if scenario == "step_nested":
    from bench import build_model
    from meshvae_hip.engine import NativeStep
    net = build_model(dev)
    net.train()
    net._prepare()
    B = 8
    x = torch.randn(B, 4998, 3, device=dev)
    y = torch.nn.functional.one_hot(torch.arange(B, device=dev) % 2, 2).float()
    eps = torch.randn(B, 16, device=dev)
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    s3 = torch.cuda.Stream(dev)
    natA = NativeStep(net, B, side_stream=s2)
    natB = NativeStep(net, B, grads=[torch.zeros_like(p) for p in net.parameters()], side_stream=s3)
    with torch.cuda.stream(main):
        natA.forward_backward(x, x, y, eps, None)
        natB.forward_backward(x, x, y, eps, None)
    torch.cuda.synchronize()


def body():
    cur = torch.cuda.current_stream(dev)
    if scenario == "plain":
        a.add_(1)
    elif scenario == "memset_fork":
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            b.zero_()                       # hipMemsetAsync -> memset node on the forked branch
        a.add_(1)
        cur.wait_stream(s1)
        a.add_(b)
    elif scenario in ("nested_fork", "nested_fork_memset"):
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            b.add_(1)
            s2.wait_stream(s1)              # a fork of a fork
            with torch.cuda.stream(s2):
                if scenario == "nested_fork_memset":
                    c.zero_()
                c.add_(2)
            b.add_(1)
            s1.wait_stream(s2)
            b.add_(c)
        a.add_(1)
        cur.wait_stream(s1)
        a.add_(b)
    elif scenario == "step_nested":
        natA.forward_backward(x, x, y, eps, None)     # chain A forks its dW lanes to s2
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            natB.forward_backward(x, x, y, eps, None)  # chain B (itself a fork) forks to s3: a fork of a fork
        cur.wait_stream(s1)
    else:
        raise SystemExit("unknown scenario")


with torch.cuda.stream(main):      # warm-up outside capture (allocations, kernel attribute calls)
    body()
torch.cuda.synchronize()
print(f"scenario {scenario}: capture", flush=True)
graph = ctypes.c_void_p()
st = ctypes.c_void_p(main.cuda_stream)
with torch.cuda.stream(main):
    call("hipStreamBeginCapture", st, 2)    # hipStreamCaptureModeRelaxed
    body()
    rc_end = call("hipStreamEndCapture", st, ctypes.byref(graph))
if rc_end != 0 or not graph.value:
    raise SystemExit("capture failed: the topology is invalid (unjoined branch?) -- see the code above")
n = ctypes.c_size_t(0)
call("hipGraphGetNodes", graph, None, ctypes.byref(n))
ne = ctypes.c_size_t(0)
call("hipGraphGetEdges", graph, None, None, ctypes.byref(ne))
print(f"  captured graph: {n.value} nodes, {ne.value} edges", flush=True)
dot = os.path.join(out_dir, f"{scenario}.dot")
call("hipGraphDebugDotPrint", graph, dot.encode(), 0)
print("  dot written:", os.path.exists(dot) and os.path.getsize(dot), flush=True)
gexec = ctypes.c_void_p()
print("  instantiating ...", flush=True)
rc = call("hipGraphInstantiate", ctypes.byref(gexec), graph, None, None, 0)
if rc == 0:
    call("hipGraphLaunch", gexec, st)
    rc2 = call("hipStreamSynchronize", st)
    print(f"scenario {scenario}: instantiate + launch ok (sync rc {rc2}); a[0]={float(a[0])} b[0]={float(b[0])} c[0]={float(c[0])}")
else:
    print(f"scenario {scenario}: hipGraphInstantiate FAILED rc={rc}")
