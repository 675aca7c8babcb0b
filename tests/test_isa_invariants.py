"""CPU: invariants of the BUILT code object that the source alone cannot guarantee — checked on the gfx950 ISA inside
libccp_gs.so (llvm-objdump), so a toolchain or source change that breaks one fails here, not silently on the GPU.

1. k_lex_wg's storer publishes progress WITHOUT draining its stores (csrc/ccp_grid_lex.hpp, lex_wg_store): after
   `s_waitcnt vmcnt(kLexPublishVmcnt)` everything older than the youngest kLexPublishVmcnt vector-memory operations is
   acknowledged, which vouches for the blocks it publishes only if the wave issues exactly kLexStoresPerBlock stores per
   8-step block — two per step plus the publication.  A compiler that merged two stores into one wider one, dropped one,
   or issued one without sc1 (write-through) would publish early or publish values still sitting in an L2: a silent
   cross-workgroup race.
2. Two workgroups of k_lex_wg share a CU only at <= 80 VGPRs (6 of their waves on one SIMD) and the kernels must not spill
   vector registers: a spill reload inside the loader's block loop waits for every prefetch in flight (round 4 measured
   +20 % per step for exactly that).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
LIB = os.path.join(ROOT, "coursecomputationalphotography_amd", "lib", "libccp_gs.so")
LLVM = "/opt/rocm/lib/llvm/bin"
HDR = os.path.join(ROOT, "coursecomputationalphotography_amd", "csrc", "ccp_grid_lex.hpp")


def _const(name):
    src = open(HDR).read()
    m = re.search(r"constexpr int %s = ([^;]+);" % name, src)
    assert m, name
    return int(eval(m.group(1), {"__builtins__": {}}))


@pytest.fixture(scope="module")
def code_objects(tmp_path_factory):
    if not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("libccp_gs.so or llvm-objdump missing")
    d = tmp_path_factory.mktemp("isa")
    so = shutil.copy(LIB, d)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, capture_output=True, cwd=d)
    objs = sorted(str(p) for p in d.iterdir() if "gfx950" in p.name)
    assert objs, "no gfx950 code object in libccp_gs.so"
    functions, notes = {}, {}
    for o in objs:
        text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", o], check=True, capture_output=True, text=True).stdout
        name = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
            if m:
                name = m.group(1)
                functions[name] = []
            elif name and line.startswith("\t"):
                functions[name].append(line.split("//")[0].strip())
        meta = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", o], check=True, capture_output=True, text=True).stdout
        cur = {}
        for line in meta.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)", line)
            if not m:
                continue
            if m.group(1) == "name" and m.group(2).startswith("_Z"):
                cur = notes.setdefault(m.group(2), cur if "name" not in cur else {})
                cur["name"] = m.group(2)
            elif m.group(1) in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size"):
                cur[m.group(1)] = int(m.group(2))
    return functions, notes


def _lex_wg(functions, depth, check, masked=False):
    want = ("k_lex_wg_masked" if masked else "k_lex_wgI") + ("" if masked else "")
    hits = [n for n in functions if ("15k_lex_wg_maskedILi%dELb%d" % (depth, check) in n if masked else "8k_lex_wgILi%dELb%d" % (depth, check) in n)]
    assert len(hits) == 1, (want, depth, check, hits)
    return hits[0], functions[hits[0]]


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("check", [0, 1])
@pytest.mark.parametrize("depth", [8, 4, 2, 1])
def test_storer_issues_exactly_its_counted_stores(code_objects, depth, check, masked):
    functions, _ = code_objects
    stores_per_block, publish_vmcnt = _const("kLexStoresPerBlock"), _const("kLexPublishVmcnt")
    assert stores_per_block == 8 * 2 + 1
    name, ins = _lex_wg(functions, depth, check, masked)
    # the publication: one s_waitcnt vmcnt(kLexPublishVmcnt), followed closely by the store of the progress word (sc1)
    pubs = [i for i, t in enumerate(ins) if t.startswith("s_waitcnt") and "vmcnt(%d)" % publish_vmcnt in t]
    assert len(pubs) == 1, (name, pubs)
    after = ins[pubs[0]:pubs[0] + 16]
    assert any(t.startswith("global_store_dword ") and "sc1" in t for t in after), after
    # Every store instruction of the kernel, accounted for one by one.  The storer's source has, per 8-step block,
    #   plain grid:  the unrolled interior branch, 8 steps x (x row: one store in the all-interior arm, one in the
    #                arm with masked lanes) + 8 x (edge values)   = 24 store instructions, each path through a step
    #                executing exactly TWO of its three; the rolled branch for the ends of a strip, 2 per iteration = 2
    #   mask grid:   one unrolled branch, 8 x 2                                                       = 16
    # all of one double, write-through (sc1).  More would mean the compiler duplicated a step (harmless, but then this
    # count has to be re-derived); fewer, or a wider store, that it merged or dropped one — the wave would issue FEWER
    # than kLexStoresPerBlock per block and publish early.  Beside them: two one-word sc1 stores (the publication and the
    # final "done") and the diagnostic stamps of CCP_GS_TRACE_FILE / the checked kernels' step sum (plain stores through
    # a scalar base, none of them in the storer's loop).
    stores = [t for t in ins if re.match(r"(global|flat|buffer)_store", t)]      # (scratch_store: spills, vouched for below)
    wide = [t for t in stores if not re.match(r"global_store_dword(x2)? ", t)]
    assert not wide, wide[:5]
    doubles = [t for t in stores if t.startswith("global_store_dwordx2 ") and "sc1" in t]
    words = [t for t in stores if t.startswith("global_store_dword ") and "sc1" in t]
    plain = [t for t in stores if "sc1" not in t]
    assert len(doubles) == (8 * 2 if masked else 8 * 3 + 2), (name, len(doubles))
    assert len(words) == 2, (name, words)
    assert len(plain) == 4 + check and all(re.match(r"global_store_dwordx2 v\d+, v\[\d+:\d+\], s\[\d+:\d+\]", t) for t in plain), plain
    # the storer loads nothing from memory: between its first store and the publication no vector load may appear on
    # the straight-line path (loads would count in vmcnt as well)
    first = min(i for i, t in enumerate(ins) if t.startswith("global_store_dwordx2 ") and "sc1" in t)
    assert not [t for t in ins[first:pubs[0]] if re.match(r"(global|flat|buffer|scratch)_load", t)], name


def test_lex_kernels_fit_two_workgroups_per_cu_and_never_spill_inside_a_step_loop(code_objects):
    functions, notes = code_objects
    seen = 0
    for name, n in notes.items():
        if "k_lex_wg" not in name:
            continue
        seen += 1
        # 6 waves of two workgroups on one SIMD: 6 x 80 <= 512 (plain and mask variant alike since round 4)
        assert n.get("vgpr_count", 0) <= 80, (name, n)
        assert n.get("group_segment_fixed_size", 0) <= 53 * 1024, (name, n)  # above ~53 KB the second workgroup is not placed (NOTES.md)
        # Spills are tolerated in a strip's prologue (once per 16,000 steps) but not where the waves step: between two
        # barriers at most a step apart, and wherever the loader has prefetches in flight (its steps issue three vector
        # loads each; its block-loop head holds the gate's polling loop).
        segs, cur = [], []
        for t in functions[name]:
            if t.startswith("s_barrier"):
                segs.append(cur)
                cur = []
            else:
                cur.append(t)
        segs.append(cur)
        for seg in segs[1:-1]:
            scratch = [t for t in seg if t.startswith("scratch_")]
            loads = [t for t in seg if re.match(r"(global|flat|buffer)_load", t)]
            lds = [t for t in seg if t.startswith("ds_")]
            # (a strip's epilogue — the checked kernels' step sum, the diagnostic stamps: plain stores — is not a loop)
            if any(t.startswith("global_store") and "sc1" not in t for t in seg):
                continue
            gate = any(t.startswith("s_sleep") for t in seg)                 # the loader's block-loop head polls and sleeps
            if (lds and len(seg) <= 64) or ((len(loads) >= 3 or gate) and len(seg) <= 120):
                assert not scratch, (name, len(seg), scratch[:3])
    assert seen == 16, seen                                                  # depths 8, 4, 2, 1 x checked / unchecked x plain / masked


def test_the_dominant_pass_keeps_two_waves_per_simd(code_objects):
    """k_fused_sweep<8, 0, 2> (the bench's dominant kernel): no spills, and at most 256 VGPRs (two waves per SIMD)."""
    _, notes = code_objects
    hits = [n for name, n in notes.items() if "13k_fused_sweepILi8ELi0ELi2E" in name]
    assert hits
    for n in hits:
        assert n.get("vgpr_spill_count", 0) == 0 and n.get("vgpr_count", 0) <= 256, n
