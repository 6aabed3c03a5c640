#!/usr/bin/env python3
"""Sample the GPU's power / clock sysfs nodes while a kernel runs back to back for a few
seconds: tells a power cap from a memory-clock change when sustained launches slow down.
usage: power_trace.py [fir|copy|add] [seconds]"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from bench import synth_stream, lowpass_taps, STREAM

what = sys.argv[1] if len(sys.argv) > 1 else "fir"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0

def nodes():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for hw in glob.glob(card + "/hwmon/hwmon*"):
            for n in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "temp3_input", "power1_cap"):
                p = os.path.join(hw, n)
                if os.path.exists(p): out[os.path.basename(card.rstrip('/device')) + ":" + n] = p
        for n in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "gpu_busy_percent", "mem_busy_percent"):
            p = os.path.join(card, n)
            if os.path.exists(p): out[n] = p
    return out

def read(p):
    try:
        s = open(p).read().strip()
    except Exception as e:
        return "ERR"
    if "\n" in s:      # pp_dpm_*: the active level carries '*'
        act = [l for l in s.split("\n") if "*" in l]
        return act[0].split(":")[1].strip().rstrip("*").strip() if act else s.replace("\n", "|")
    return s

N = nodes()
print("nodes:", list(N.keys()), flush=True)
ctx = ap.Context(0)
n = STREAM
ins = [ctx.vec(synth_stream(815 + i, n)) for i in range(4)]
outs = [ctx.empty(n) for _ in range(4)]
fir = ap.Fir(ctx, lowpass_taps(), 2048)
def launch(i):
    if what == "fir": fir.filter(ins[i % 4], out=outs[i % 4])
    elif what == "copy": outs[i % 4].vec_clone(ins[i % 4])
    else: outs[i % 4].vec_add(ins[i % 4])
samples = []
stop = False
def sampler():
    t0 = time.time()
    while not stop:
        samples.append((time.time() - t0, {k: read(p) for k, p in N.items()}))
        time.sleep(0.02)
th = threading.Thread(target=sampler); th.start()
time.sleep(0.3)                         # idle baseline
e0, e1 = ctx.event(), ctx.event()
t_start = time.time(); k = 0; marks = []
while time.time() - t_start < secs:
    e0.record()
    for i in range(200): launch(k + i)
    e1.record(); ctx.sync(); k += 200
    marks.append((time.time() - t_start, e0.elapsed_ms(e1) / 200 * 1e3))
time.sleep(0.3)
stop = True; th.join()
print(f"{what}: per-launch us over time:", " ".join(f"{t:.2f}s:{us:.1f}" for t, us in marks[::max(1, len(marks)//16)]))
keys = list(N.keys())
print("t      " + "  ".join(keys))
for t, d in samples[::max(1, len(samples)//40)]:
    print(f"{t:5.2f}  " + "  ".join(str(d[k]) for k in keys))
