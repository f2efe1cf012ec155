"""The build's ISA lint (tinychatengine_amd/isa_lint.py, round 6): RULE 1 -- no packed-f32 arithmetic whose low half selects a high source register -- is what stands
between the library and round 5's lost accumulator lanes (profiles/r6/pk_lost_lanes_rule.md).  Held here without a GPU: the lint sees the form when it is there (a
six-line kernel holding it through inline assembly, cross-compiled for gfx950), passes the forms that measured clean, and every object of the product library is clean."""
import os
import subprocess

import pytest

from tinychatengine_amd import build as tce_build
from tinychatengine_amd import isa_lint

_KERNEL = r"""
#include <hip/hip_runtime.h>
typedef float float2_t __attribute__((ext_vector_type(2)));
extern "C" __global__ void probe(float2_t *p) {
    float2_t a = p[threadIdx.x], e = p[threadIdx.x + 64], d;
    asm volatile("%s %%0, %%1, %%2 %s" : "=v"(d) : "v"(a), "v"(e));
    p[threadIdx.x] = d;
}
"""


def _compile(tmp_path, name, op, sel):
    src = tmp_path / f"{name}.hip"
    src.write_text(_KERNEL % (op, sel))
    obj = tmp_path / f"{name}.o"
    subprocess.check_call([tce_build._hipcc(), "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", str(obj)])
    return str(obj)


@pytest.mark.parametrize("op,sel,bad", [("v_pk_mul_f32", "op_sel:[0,1]", True), ("v_pk_add_f32", "op_sel:[0,1]", True), ("v_pk_mul_f32", "op_sel:[1,0]", True),
                                        ("v_pk_mul_f32", "op_sel_hi:[1,0]", False), ("v_pk_mul_f32", "", False), ("v_pk_mov_b32", "op_sel:[0,1]", False)])
def test_rule_1_sees_the_form(tmp_path, op, sel, bad):
    names, viol = isa_lint.lint_object(_compile(tmp_path, "k", op, sel))
    assert names == ["probe"]
    assert bool(viol) == bad, (op, sel, viol)
    if bad:
        assert viol[0][0] == "probe" and viol[0][2] == 1 and op in viol[0][1]
        assert "RULE 1" in isa_lint.format_violations("k.o", viol)


def test_rule_1_parses_the_three_operand_form():
    assert isa_lint.rule1_violations(["v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]", "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0,1]",
                                      "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,0] op_sel_hi:[1,0]", "v_mul_f32_e32 v0, v1, v2"]) == ["v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]"]


def test_every_object_of_the_library_is_clean():
    tce_build.build()
    objs = sorted(f for f in os.listdir(tce_build.LIB_DIR) if f.endswith(".o"))
    assert len(objs) >= len(tce_build.HIP_SOURCES)
    total = 0
    for o in objs:
        names, viol = isa_lint.lint_object(os.path.join(tce_build.LIB_DIR, o))
        assert not viol, isa_lint.format_violations(o, viol)
        total += len(names)
    assert total > 300  # (the disassembly was really read: ~590 kernels today)
