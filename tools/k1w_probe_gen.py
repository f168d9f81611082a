"""Timing variants of the generated K1 count loop for tools/k1w_probe.hip (never part of the library)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "rankcompv3.jl_amd", "csrc"))
from gen_k1_loop import Loop

VARIANTS = [("base", ()), ("wait4", ("wait4",)), ("nowaitvm", ("nowait_vm",)), ("noreload", ("noreload",)), ("nowaitlds", ("nowait_lds",)),
            ("nowait", ("nowait_lds", "nowait_vm")), ("nolds", ("nolds",)), ("nopop", ("nopop",)), ("lshl", ("lshl",)), ("spread", ("spread",)),
            ("valuonly", ("noreload", "nolds")), ("bitoponly", ("noreload", "nolds", "nopop"))]
print("typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));")
for name, opts in VARIANTS:
    lp = Loop(15, False, "probe_" + name, opts=opts)
    lp.generate()
    print(lp.cxx())
print("typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));")
for name, opts in (("half", ()), ("half_noreload", ("noreload",))):
    lp = Loop(15, False, "probe16_" + name, opts=opts, ri=16)
    lp.generate()
    print(lp.cxx())
print("#define PROBE_VARIANTS(X) " + " ".join(f'X({i}, probe_{n}, "{n}")' for i, (n, _) in enumerate(VARIANTS)))
