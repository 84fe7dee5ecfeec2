import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import voxtral_c_amd as v
from conftest import model_dir
d = model_dir("full")
for i in range(4):
    t0 = time.time()
    m = v.Model(d)
    t1 = time.time()
    m.close()
    print("load %.3f s" % (t1 - t0), flush=True)
