import os
import sys
os.environ.setdefault("DACAPO_AMD_HOOKS", "1")  # seeded keys: the hooks build (csrc/test_hooks.hip)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dacapo_amd import runner, hevm_asm as ha
def p(*a): print(*a, flush=True)
hevm = runner.HEVM(seed=1, logN=13, num_primes=7); p("vm ok")
rng = np.random.default_rng(100); img = rng.uniform(0, 1, 4096)
b = ha.sobel_filter(img, slots=hevm.slots, init_level=6)
cst, hv, info = b.assemble(); p(info)
hevm.load_mem(cst, hv); p("loaded")
hevm.setInput(0, img); p("encrypted")
hevm.run(); p("run1")
hevm.run(); p("run2")
res = hevm.getOutput(); p("rms", np.sqrt(np.mean((res[0]-b.expected()[0])**2)))
