import os, sys
os.environ.setdefault("DACAPO_AMD_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dacapo_amd import hevm_asm as ha, runner
logN, K, ks = 12, 9, 3
slots = 1 << (logN - 1)
rng = np.random.default_rng(23)
b = ha.Builder(slots=slots, init_level=K - ks, policy="lazy", boot_level=K - ks, shadow=True)
x, y = b.input(rng.uniform(-1, 1, slots)), b.input(rng.uniform(-1, 1, slots))
pt = lambda: rng.uniform(-1, 1, slots)
baby = b.rotate(y, 3)
bsgs = b.add(b.add(b.mul_plain(baby, pt()), b.rotate(b.mul_plain(baby, pt()), 8)), b.mul_plain(b.rotate(x, 5), pt()))
b.output(b.finish(bsgs))
cst, hv, _ = b.assemble()
names = {0: "enc", 1: "rot", 2: "neg", 3: "rs", 4: "ms", 6: "addcc", 7: "addcp", 8: "mulcc", 9: "mulcp", 10: "boot"}
for i, (o, d, l, r) in enumerate(ha.unpack_hevm(hv)["ops"].tolist()):
    print(i, names.get(o, o), d, l, r)
hevm = runner.HEVM(seed=9, logN=logN, num_primes=K, ks_special=ks, vm_options={"plan": 1, "plan_graph": 0, "hyb_lazy_sum": 1, "hyb_double_hoist": 1, "trace": 2})
hevm.addRotationKeys([3, 5])
hevm.load_mem(cst, hv)
for i, a in enumerate(b.args):
    hevm.setInput(i, a.plain)
hevm.run()
print(hevm.lazy_groups())
