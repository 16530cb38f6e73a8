#!/usr/bin/env python3
"""Why does rr_legacy_randn take 2 ms alone and 4.5-5 ms inside fit()?  Same call: back to back; with 5 ms idle gaps; from a
worker thread while the main thread (a) sleeps in C, (b) runs Python bytecode, (c) sits in a GPU synchronisation."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd import _hip
n = 10 * 50 * 2048
rs = np.random.RandomState(0)
def med(f, reps=30):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))
call = lambda: _hip.legacy_randn(rs, n, np.float32, threads=2)
print("back to back            %.2f ms" % med(call))
def gap():
    time.sleep(0.005); 
ts = []
for _ in range(30):
    time.sleep(0.005); t = time.perf_counter(); call(); ts.append(time.perf_counter() - t)
print("5 ms idle gaps          %.2f ms" % (1e3 * np.median(ts)))
def in_thread(main_work, label):
    out, stop = [], threading.Event()
    def w():
        for _ in range(30):
            time.sleep(0.003); t = time.perf_counter(); call(); out.append(time.perf_counter() - t)
        stop.set()
    th = threading.Thread(target=w); th.start()
    while not stop.is_set(): main_work()
    th.join()
    print("%-23s %.2f ms" % (label, 1e3 * np.median(out)))
in_thread(lambda: time.sleep(0.002), "thread, main sleeps")
x = np.arange(200.0)
def py():
    s = 0.0
    for i in range(20000): s += i * 0.5
in_thread(py, "thread, main in Python")
try:
    dev = _hip.get_device()
    buf = dev.zeros(1 << 28)
    def gpu():
        for _ in range(8): dev.memset(buf)
        dev.sync()
    in_thread(gpu, "thread, main in GPU sync")
except Exception as e:
    print("no gpu:", e)
