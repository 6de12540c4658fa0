"""tools/valu_sched.py (the post-pass scheduler of profiles/r03_lighting_diet.txt box F, a measurement tool, not part of the product
build): its output must be a permutation of its input in which every register dependency keeps its direction, nothing crosses a
barrier, and fixed (scalar / memory) instructions keep their order."""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

SNIPPET = """\t.text
_Z22lighting_stream_kernelv:
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
\tv_cvt_f32_f16_e32 v1, v10
\tv_cvt_f32_f16_e32 v2, v11
\tv_mul_f32_e32 v3, v1, v1
\tv_fmac_f32_e32 v3, v2, v2
\tv_rsq_f32_e32 v3, v3
\ts_add_i32 s4, s4, 1
\tv_floor_f32_e32 v5, v12
\tv_floor_f32_e32 v6, v13
\tv_mul_f32_e32 v7, v1, v3
\tv_mul_f32_e32 v8, v2, v3
\tds_read_b32 v20, v14
\tv_sub_f32_e32 v9, v12, v5
\tv_sub_f32_e32 v15, v13, v6
\tv_fma_f32 v16, s8, v9, v15
\ts_waitcnt lgkmcnt(0)
\tv_add_f32_e32 v21, v20, v16
\tv_mul_f32_e32 v22, v21, v7
\tv_mul_f32_e32 v23, v21, v8
\ts_cbranch_scc1 .LBB0_1
.Lfunc_end0:
"""


def _instructions(text):
    return [l.strip() for l in text.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]


def test_output_is_a_dependency_preserving_permutation(tmp_path):
    import valu_sched as V
    src, dst = tmp_path / "a.s", tmp_path / "b.s"
    src.write_text(SNIPPET)
    sys.argv = ["valu_sched.py", str(src), str(dst), "--kernels", "lighting_stream_kernel"]
    V.main()
    a, b = _instructions(SNIPPET), _instructions(dst.read_text())
    assert sorted(a) == sorted(b), "a permutation of the same instructions"
    assert a != b, "the snippet has pairs to form (two floors in a row, dependent neighbours)"
    pos = {t: i for i, t in enumerate(b)}
    # nothing crosses the wait or the branch
    wait_a, wait_b = a.index("s_waitcnt lgkmcnt(0)"), b.index("s_waitcnt lgkmcnt(0)")
    assert wait_a == wait_b and set(a[:wait_a]) == set(b[:wait_b])
    assert b[-1].startswith("s_cbranch")
    # scalar and memory instructions keep their relative order
    fixed = [t for t in a if t.startswith(("s_", "ds_"))]
    assert [t for t in b if t.startswith(("s_", "ds_"))] == fixed

    def regs(s):
        return set(re.findall(r"\b[vs]\d+\b", s))
    for i, x in enumerate(a):
        wx = regs(x.split(",")[0])
        for y in a[i + 1:]:
            ops = y.split(",")
            ry, wy = regs(",".join(ops[1:])) | (regs(ops[0]) if "fmac" in y else set()), regs(ops[0])
            if (wx & ry) or (wx & wy) or (regs(",".join(x.split(",")[1:])) & wy):
                assert pos[x] < pos[y], f"dependency reversed: {x!r} -> {y!r}"
    # the consumer of the transcendental's result is not directly behind it
    k = b.index("v_rsq_f32_e32 v3, v3")
    assert "v3" not in regs(",".join(b[k + 1].split(",")[1:]))
