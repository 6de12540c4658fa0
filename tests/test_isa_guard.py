"""Static guard on the shipped gfx950 code of the streaming lighting kernel (CPU test: no GPU needed).

The kernel's tile prefetch is LDS-DMA (global_load_lds_dwordx4), which hipcc's s_waitcnt bookkeeping does not see: the
ordering protocol of csrc/lighting.hip ("in iteration t the DMA for tile t+2 is issued where no load of hipcc's own is
pending; an explicit vmcnt(0) in front of it retires the DMA issued one iteration earlier") is otherwise verified only by the
parity tests. A compiler bump that moves a wait or starts spilling would break it silently, so this test unbundles the
code object from libur_hotpath.so and checks, for lighting_stream_kernel<FUSED, shadows, IRR_LDS, 16> and its siblings:
  * no scratch (private_segment_fixed_size == 0), no VGPR spills, <= 128 VGPRs (4 waves per SIMD), at most two spilled SGPRs
    and none of their lane traffic (v_writelane / v_readlane) inside the persistent loop;
  * on every control-flow path that leads to a global_load_lds_dwordx4 of the loop there is an `s_waitcnt vmcnt(0)` with no
    VGPR-destination vector load between it and the DMA (so neither a gather result nor the DMA of the previous iteration
    is outstanding when it issues), and no path goes round the loop to the same DMA without such a wait;
  * the loop issues its DMAs in pairs (A|B and HDR|C|depth of one tile) and contains no barrier."""
import re
import struct
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "unclerenderer_amd" / "csrc" / "_build" / "libur_hotpath.so"
LLVM = Path("/opt/rocm/lib/llvm/bin")
HOT = "lighting_stream_kernelILi2ELb1ELb1ELi16ELb0E"  # MODE = FUSED, SHADOWS, IRR_LDS, 16 waves per workgroup, pieces by the last wave: the bench kernel


def _code_objects(tmp_path: Path) -> list[Path]:
    fat = tmp_path / "fat.bin"
    subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(LIB), str(tmp_path / "discard.so")], check=True)
    data = fat.read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], 0
    while (i := data.find(magic, pos)) >= 0:
        (n,) = struct.unpack_from("<Q", data, i + 24)
        o = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode()
            o += tl
            if "gfx950" in triple and size:
                p = tmp_path / f"co_{len(out)}.elf"
                p.write_bytes(data[i + off:i + off + size])
                out.append(p)
        pos = i + len(magic)
    return out


@pytest.fixture(scope="module")
def lighting_co(tmp_path_factory, urlib):
    if not (LLVM / "llvm-objdump").exists():
        pytest.skip("llvm tools not found")
    tmp = tmp_path_factory.mktemp("isa")
    for co in _code_objects(tmp):
        syms = subprocess.run([str(LLVM / "llvm-readelf"), "-s", "--wide", str(co)], capture_output=True, text=True, check=True).stdout
        if HOT in syms:
            return co
    pytest.fail("the streaming lighting kernel is not in libur_hotpath.so")


def _kernel_metadata(co: Path) -> dict[str, dict[str, int]]:
    notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True, check=True).stdout
    kernels, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'\"")
        if k == "name" and v.startswith("_Z") and not v.endswith(".kd"):
            cur = kernels.setdefault(v, {})
        elif cur is not None and k in ("private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count", "vgpr_count", "sgpr_count", "group_segment_fixed_size") and v.isdigit():
            cur[k] = int(v)
    return kernels


def test_streaming_kernels_have_no_scratch_and_no_spills(lighting_co):
    meta = {k: v for k, v in _kernel_metadata(lighting_co).items() if "lighting_stream_kernel" in k}
    assert len(meta) == 32, sorted(meta)  # 2 modes x shadows x IRR_LDS x {12, 16} waves x {riding HZB pieces by the last wave, by every wave}
    hot = next(v for k, v in meta.items() if HOT in k)
    # (an SGPR spill is a v_writelane / v_readlane pair, not scratch memory; the two the bench kernel has are written ahead of
    # the loop and read behind it: test_no_spill_traffic_inside_the_loop)
    assert hot["private_segment_fixed_size"] == 0 and hot["vgpr_spill_count"] == 0 and hot["sgpr_spill_count"] <= 8, hot
    assert hot["vgpr_count"] <= 128, hot  # 4 waves per SIMD need <= 128
    for name, m in meta.items():
        assert m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0, (name, m)
        if "Li16ELb" in name:  # the shipped configuration (UR_OPT_LIGHTING_WAVES_PER_WG = 12 is a diagnostic one)
            assert m["sgpr_spill_count"] <= 8 and m["vgpr_count"] <= 128, (name, m)  # (what the loop sees of them: test_no_spill_traffic_inside_the_loop)


def _disassemble(co: Path, symbol_part: str) -> tuple[list[str], dict[str, int]]:
    """Instructions of one kernel in program order and the index each branch label (L123) refers to."""
    text = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", "--symbolize-operands", str(co)], capture_output=True, text=True,
                          check=True).stdout
    ins, labels, on = [], {}, False
    for line in text.splitlines():
        if re.match(r"^[0-9a-f]+ <_Z", line):
            on = symbol_part in line
            continue
        if not on or not line.strip():
            continue
        m = re.match(r"^[0-9a-f]+ <(L\d+)>:", line.strip())
        if m:
            labels[m.group(1)] = len(ins)
            continue
        ins.append(line.split("//")[0].strip())
    assert ins, symbol_part
    return ins, labels


def _predecessors(ins: list[str], labels: dict[str, int]) -> list[list[int]]:
    pred = [[] for _ in ins]
    for i, t in enumerate(ins):
        op = t.split()[0]
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = labels.get(t.split()[-1])
            assert tgt is not None, t
            if tgt < len(ins):
                pred[tgt].append(i)
        if i + 1 < len(ins) and op not in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pred[i + 1].append(i)
    return pred


def _reach(adj: list[list[int]], start: int) -> set[int]:
    seen, stack = set(), list(adj[start])
    while stack:
        j = stack.pop()
        if j not in seen:
            seen.add(j)
            stack.extend(adj[j])
    return seen


def _cycle_members(pred: list[list[int]], n: int) -> set[int]:
    """Instructions that can reach themselves again."""
    return {i for i in range(n) if i in _reach(pred, i)} if n < 200 else _cycle_members_fast(pred, n)


def _cycle_members_fast(pred: list[list[int]], n: int) -> set[int]:
    succ = [[] for _ in range(n)]
    for i, ps in enumerate(pred):
        for p in ps:
            succ[p].append(i)
    # Kosaraju: components of size > 1 (or with a self edge) lie on cycles
    order, seen = [], [False] * n
    for r in range(n):
        if seen[r]:
            continue
        stack = [(r, 0)]
        seen[r] = True
        while stack:
            v, k = stack.pop()
            if k < len(succ[v]):
                stack.append((v, k + 1))
                w = succ[v][k]
                if not seen[w]:
                    seen[w] = True
                    stack.append((w, 0))
            else:
                order.append(v)
    comp, out = [-1] * n, set()
    for r in reversed(order):
        if comp[r] != -1:
            continue
        members, stack = [], [r]
        comp[r] = r
        while stack:
            v = stack.pop()
            members.append(v)
            for w in pred[v]:
                if comp[w] == -1:
                    comp[w] = r
                    stack.append(w)
        if len(members) > 1:
            out.update(members)
    _cycle_members_fast.comp = comp
    return out


def _component_of(pred, node, on_cycle) -> set[int]:
    comp = _cycle_members_fast.comp
    return {i for i in on_cycle if comp[i] == comp[node]}


@pytest.mark.parametrize("kernel", [HOT, "lighting_stream_kernelILi2ELb1ELb1ELi16ELb1E", "lighting_stream_kernelILi2ELb1ELb0ELi16ELb0E", "lighting_stream_kernelILi0ELb1ELb1ELi16ELb0E",
                                    "lighting_stream_kernelILi2ELb0ELb1ELi16ELb0E"])
def test_dma_ordering_protocol_in_the_isa(lighting_co, kernel):
    ins, labels = _disassemble(lighting_co, kernel)
    _check_protocol(ins, labels, kernel)


def test_the_guard_notices_a_missing_wait(lighting_co):
    """The walk is not vacuous: with the vmcnt(0) in front of the loop's DMAs deleted from the listing it must object."""
    ins, labels = _disassemble(lighting_co, HOT)
    dma = [i for i, t in enumerate(ins) if _is_tile_dma(t)]
    cut = list(ins)
    for a in dma:
        for j in range(a - 1, max(a - 120, 0), -1):
            if cut[j].startswith("s_waitcnt") and "vmcnt(0)" in cut[j]:
                cut[j] = "s_nop 0"
                break
    with pytest.raises(AssertionError):
        _check_protocol(cut, labels, HOT)


def _is_tile_dma(t: str) -> bool:
    """The two DMA instructions of a G-buffer tile (tile_dma_at: `... nt` then `... offset:1024`). The prologue's one-off DMA of the
    cube's small mips into LDS (dma16: neither modifier) is waited for ahead of the loop and is not part of the protocol."""
    return t.startswith("global_load_lds_dwordx4") and (" nt" in t or "offset:1024" in t)


def _check_protocol(ins, labels, kernel):
    pred = _predecessors(ins, labels)
    barriers = [i for i, t in enumerate(ins) if t.startswith("s_barrier")]
    dma = [i for i, t in enumerate(ins) if _is_tile_dma(t)]
    assert barriers and len(dma) >= 8 and len(dma) % 2 == 0
    for a, b in zip(dma[0::2], dma[1::2]):
        assert b == a + 1, "the two DMA instructions of a tile (A|B and HDR|C|depth) are issued back to back"
    is_vgpr_load = lambda t: re.match(r"global_load_(dword|dwordx2|dwordx3|dwordx4|ushort|ubyte|short|sbyte|sshort)\b", t) is not None
    # a loop DMA = one on a cycle of the control-flow graph (the prologue's static tiles are issued once, with nothing of the
    # loop in flight): the persistent loop is the strongly connected component that holds DMAs
    on_cycle = _cycle_members(pred, len(ins))
    loop_dma = [i for i in dma[0::2] if i in on_cycle]
    assert len(loop_dma) >= 2, dma
    loop = _component_of(pred, loop_dma[0], on_cycle)
    assert all(i in loop for i in loop_dma)
    assert not any(b in loop for b in barriers), "no barrier inside the persistent loop"
    head, tail = min(loop), max(loop)
    assert sum(is_vgpr_load(ins[i]) for i in loop) >= 4  # there are gathers in the loop: the walk below is not vacuous
    for a in loop_dma:
        # walk EVERY backward path from the DMA until an `s_waitcnt vmcnt(0)`: no gather result and no older DMA may be met first
        stack, seen, waits = list(pred[a]), set(), 0
        while stack:
            j = stack.pop()
            if j in seen:
                continue
            seen.add(j)
            t = ins[j]
            if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                waits += 1
                continue
            assert not is_vgpr_load(t), f"{kernel}: `{t}` may be in flight when the tile DMA at instruction {a} issues"
            # (the DMA pair of the OTHER arm of the whole-tile / partial-tile choice lies on a statically possible path: hipcc joins
            # the arms through a flag register; it is skipped, the walk goes on to the wait in front of both arms)
            assert j != a, f"{kernel}: a path around the loop reaches the DMA at {a} again without a vmcnt(0)"
            assert len(seen) < 400, f"{kernel}: no s_waitcnt vmcnt(0) near the DMA at instruction {a}"
            stack.extend(pred[j])
        assert waits >= 1


RIDE_ALL = "lighting_stream_kernelILi2ELb1ELb1ELi16ELb1E"  # the same kernel with every wave walking Build HZB pieces (short bands)


@pytest.mark.parametrize("kernel", [HOT, RIDE_ALL, "lighting_stream_kernelILi0ELb1ELb1ELi16ELb0E"])
def test_no_spill_traffic_inside_the_loop(lighting_co, kernel):
    """The persistent loop = the strongly connected component of the control-flow graph that holds the tile DMAs (the shading body,
    its all-sky sub-loop and, laid out behind it, the cold run-time claim path dyn_claim, which jumps back in). No SGPR-spill STORE
    (v_writelane) and no scratch access anywhere on it; spill RELOADS (v_readlane) only the few the cold claim path needs - the
    bench kernel has eight there and none in the shading body; the every-wave-walks variant keeps one extra scalar of its pre-loop
    code across the loop (two more reloads). A compiler that starts spilling in the shading body trips the count."""
    ins, labels = _disassemble(lighting_co, kernel)
    pred = _predecessors(ins, labels)
    dma = [i for i, t in enumerate(ins) if _is_tile_dma(t)]
    on_cycle = _cycle_members(pred, len(ins))
    loop_dma = [i for i in dma[0::2] if i in on_cycle]
    assert len(loop_dma) >= 2
    loop = _component_of(pred, loop_dma[0], on_cycle)
    assert len(loop) > 600, len(loop)  # the shading loop, not a sub-loop of it
    bad = [ins[i] for i in sorted(loop) if ins[i].startswith(("v_writelane", "scratch_", "buffer_store", "buffer_load"))]
    assert not bad, bad[:8]
    reloads = [(i, ins[i]) for i in sorted(loop) if ins[i].startswith("v_readlane")]
    assert len(reloads) <= (10 if kernel == RIDE_ALL else 8), reloads
    # ... and none of them in front of the shading body's gathers: the reloads sit behind the last cube / shadow gather of the listing
    gathers = [i for i in sorted(loop) if re.match(r"global_load_dwordx3\b", ins[i])]
    assert gathers
    if kernel != RIDE_ALL:
        assert all(i > max(gathers) for i, _ in reloads), (reloads, max(gathers))


def test_the_product_kernel_source_carries_no_variant_switches():
    """The measured-and-rejected structures of rounds 1-2 (producer/consumer wave specialisation, ablation switches, in-kernel
    stamps, store / DMA cache-policy alternatives) live in git history and are rebuilt from there by tools/build_variants.py
    (--base r02); the product translation unit has one loop, and this guard checks exactly that loop."""
    src = (ROOT / "unclerenderer_amd" / "csrc" / "lighting.hip").read_text()
    for switch in ("UR_LOADER_WAVE", "UR_ABLATE", "UR_STAMP", "UR_HDR_STORE", "UR_DMA_NT", "UR_RIDE_RELEASE_FENCE"):
        assert switch not in src, f"{switch} is back in the product source"
