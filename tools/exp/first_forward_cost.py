import sys, time; sys.path.insert(0,'.')
import numpy as np
from davo_amd import Engine, synth, parse_version, FLAGSHIP_VERSION
cfg=parse_version(FLAGSHIP_VERSION); w=synth.make_weights(cfg)
d=synth.make_inputs(8,128,416)
for k in range(3):
    t0=time.perf_counter(); e=Engine(cfg,128,416,64); t1=time.perf_counter(); e.load_weights(w); t2=time.perf_counter()
    e.calibrate(*d); t3=time.perf_counter(); e.forward(*d); t4=time.perf_counter()
    print("engine %d: create %.3f load_weights %.3f calibrate(first forward) %.3f next forward %.4f" % (k, t1-t0, t2-t1, t3-t2, t4-t3), flush=True)
    e.close()
