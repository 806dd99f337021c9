#!/usr/bin/env python3
"""How long does the host take to notice that a short kernel has ended?  One tiny kernel + stream / event / device
synchronise, under the host-wait settings HIP offers (development probe behind bench.py's choice).
  python tools/sync_latency.py            # runs every variant in a child process
"""
import os
import subprocess
import sys
import time

VARIANTS = {
    "default": {},
    "spin_flag": {"PROBE_FLAGS": "1"},          # hipSetDeviceFlags(hipDeviceScheduleSpin) before the context exists
    "yield_flag": {"PROBE_FLAGS": "2"},
    "blocking_flag": {"PROBE_FLAGS": "4"},
    "active_wait_1000us": {"ROC_ACTIVE_WAIT_TIMEOUT": "1000"},
    "active_wait_100000us": {"ROC_ACTIVE_WAIT_TIMEOUT": "100000"},
}


def child():
    import ctypes as C
    flags = int(os.environ.get("PROBE_FLAGS", "0"))
    if flags:
        hip = C.CDLL("libamdhip64.so")
        print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(C.c_uint(flags)))
    import torch
    dev = torch.device("cuda:0")
    x = torch.zeros(1024, device=dev)
    s = torch.cuda.current_stream()
    for _ in range(50):
        x.add_(1)
    torch.cuda.synchronize()
    out = {}
    for name, wait in (("device_sync", lambda ev: torch.cuda.synchronize()), ("stream_sync", lambda ev: s.synchronize()),
                       ("event_sync", lambda ev: ev.synchronize())):
        ts = []
        for _ in range(300):
            ev = torch.cuda.Event()
            t0 = time.perf_counter()
            x.add_(1)
            ev.record(s)
            wait(ev)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        out[name] = (ts[len(ts) // 2] * 1e6, ts[int(len(ts) * 0.9)] * 1e6)
    print({k: f"median {a:.1f} us, p90 {b:.1f} us" for k, (a, b) in out.items()})


if __name__ == "__main__":
    if os.environ.get("PROBE_CHILD"):
        child()
        sys.exit(0)
    for name, env in VARIANTS.items():
        r = subprocess.run([sys.executable, __file__], env=dict(os.environ, PROBE_CHILD="1", **env), capture_output=True, text=True)
        print(name, (r.stdout.strip() or r.stderr.strip()[-300:]).replace("\n", " | "), flush=True)
