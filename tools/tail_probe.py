"""Host-side latencies around the step launch (one MI355X): synchronisation flavours, eager launch rate, the Python
wrapper's per-call cost.  Run on the GPU box: python tools/tail_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from control_pcgrl_amd import VecPcgrlEnv
dev = torch.device("cuda:0")
env = VecPcgrlEnv("binary", "narrow", (16, 16), 4096, seeds=np.arange(4096), auto_reset=True)
env.reset()
acts = torch.randint(0, 2, (64, 4096), device=dev, dtype=torch.int32)
sptr = torch.cuda.current_stream(dev).cuda_stream
ep_host = torch.zeros(5, dtype=torch.float64).pin_memory()
for k in range(200):
    env.step_raw(acts[k % 64].data_ptr(), sptr)
torch.cuda.synchronize()
def t(f, n=200):
    xs = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); xs.append(time.perf_counter() - t0)
    xs.sort(); return xs[len(xs)//2] * 1e6
print("sync idle            %.1f us" % t(lambda: torch.cuda.synchronize(dev)))
print("stream sync idle     %.1f us" % t(lambda: torch.cuda.current_stream(dev).synchronize()))
def one():
    env.step_raw(acts[0].data_ptr(), sptr); torch.cuda.synchronize(dev)
print("1 step + sync        %.1f us" % t(one))
def one_s():
    env.step_raw(acts[0].data_ptr(), sptr); torch.cuda.current_stream(dev).synchronize()
print("1 step + stream sync %.1f us" % t(one_s))
def red():
    env._L.pcgrl_reduce_episodes(env._h, ep_host.data_ptr(), 1, sptr); torch.cuda.synchronize(dev)
print("reduce + sync        %.1f us" % t(red))
def twenty():
    for k in range(20): env.step_raw(acts[k].data_ptr(), sptr)
    torch.cuda.synchronize(dev)
print("20 steps + sync      %.1f us" % t(twenty, 50))
def twenty_r():
    for k in range(20): env.step_raw(acts[k].data_ptr(), sptr)
    env._L.pcgrl_reduce_episodes(env._h, ep_host.data_ptr(), 1, sptr); torch.cuda.synchronize(dev)
print("20 steps+reduce+sync %.1f us" % t(twenty_r, 50))
ev = torch.cuda.Event()
def twenty_e():
    for k in range(20): env.step_raw(acts[k].data_ptr(), sptr)
    ev.record(); ev.synchronize()
print("20 steps + event sync %.1f us" % t(twenty_e, 50))
def launch_only():
    for k in range(20): env.step_raw(acts[k].data_ptr(), sptr)
print("20 launches (host)   %.1f us" % t(launch_only, 50)); torch.cuda.synchronize()
# the Python wrapper a policy-in-the-loop user calls (tensor checks + ctypes + launch), host side only
a32 = acts[0].contiguous()
def wrapped():
    for k in range(20): env.step(a32)
print("20 x VecPcgrlEnv.step (host) %.1f us" % t(wrapped, 50)); torch.cuda.synchronize()
def wrapped_sync():
    for k in range(20): env.step(a32)
    torch.cuda.synchronize(dev)
print("20 x VecPcgrlEnv.step + sync %.1f us" % t(wrapped_sync, 50))
