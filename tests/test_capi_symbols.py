"""CPU-side checks of the C-ABI boundary: the in-tree library loads, exports EVERY symbol that
include/glnn_hip.h declares, the Python binding table matches the header, and ops refuse CPU tensors
(no silent fallback).  No compute call is made here."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "glnn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"GLNN_API\s+[\w\s\*]+?\b(glnn_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    lib_path = ge.build()
    h = ctypes.CDLL(lib_path)
    fns = header_functions()
    assert len(fns) >= 15
    for name in fns:
        assert hasattr(h, name), f"{name} declared in include/glnn_hip.h but not exported"
    h.glnn_abi_version.restype = ctypes.c_int
    assert h.glnn_abi_version() == 12
    h.glnn_last_error.restype = ctypes.c_char_p
    assert h.glnn_last_error() is not None


def test_binding_table_matches_header():
    import glnn_amd
    from glnn_amd import _lib
    fns = header_functions()
    for name, argtypes in _lib.SIGNATURES.items():
        assert name in fns, name
        assert len(argtypes) == fns[name], f"{name}: binding has {len(argtypes)} args, header {fns[name]}"
    assert set(fns) - set(_lib.SIGNATURES) == {"glnn_last_error", "glnn_reload_options"}
    glnn_amd.lib()


def test_invalid_arguments_are_reported_not_crashed():
    from glnn_amd import _lib
    h = _lib.lib()
    rc = h.glnn_spmm_csr_f32(None, None, 4, 4, None, 4, 4, 0, None, None, None, 0, None, None, None, 0, None, 4, None)
    assert rc == -1 and b"null pointer" in h.glnn_last_error()
    assert h.glnn_spmm_csr_f32(None, None, 0, 0, None, 4, 4, 0, None, None, None, 0, None, None, None, 0, None, 4, None) == 0   # empty: no-op
    rc = h.glnn_gemm_f32(None, 4, None, None, None, 0.0, 0, 4, 4, None, 4, 0, 4, None, None, None, 0, None, 4, None, 0, None)
    assert rc == -1


def test_ops_refuse_cpu_tensors():
    from glnn_amd import GlnnError, ops
    x = torch.zeros(4, 4)
    with pytest.raises(GlnnError):
        ops.gemm(x, x)
    with pytest.raises(GlnnError):
        ops.log_softmax(x)
    with pytest.raises(GlnnError):
        ops.spmm(torch.zeros(5, dtype=torch.int64), torch.zeros(1, dtype=torch.int32), x, 4, ops.AGG_SUM)


def test_pipelined_gemm_loops_contain_no_vector_alu_and_no_register_copies(tmp_path):
    """gemm_kernel_pipe / gemm_tn_kernel_pipe issue their operand loads from inline asm: the compiler does not know those loads are
    in flight, so any register copy (or other VALU instruction) it placed on a staging / fragment register between the prologue
    barrier and the final drain would read registers that have not been written yet.  Checked on the ISA hipcc generates for gfx950
    (also the performance contract: no vector-ALU work between the MFMAs of the main loop body).  The only VALU work in that region
    is the out-of-line block that moves the accumulators into the block totals every 8 k-tiles (round 4): it may touch nothing but
    accumulator / total registers.
    gemm_tn_kernel_pipe_ax (round 5: the A operand is alpha a + beta z + gamma, evaluated on the staged pieces) is the one exception, and a
    stated one: its loop holds exactly 32 v_pk_fma_f32 (4 per A piece, 4 pieces, 2 k-tiles) and no other vector-ALU instruction, and each
    of them reads staged registers only behind the hand-placed s_waitcnt vmcnt that covers their loads.  (Round 6: its prologue may also
    hold one loop -- the constants made from the tile partials -- which must end before the first asm load is issued.)"""
    import re, shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "graphless-neural-networks_amd", "csrc", "gemm.hip")
    out = tmp_path / "gemm.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", f"-I{ROOT}/include",
                    f"-I{ROOT}/graphless-neural-networks_amd/csrc", "-S", "--cuda-device-only", "-o", str(out), src],
                   check=True, capture_output=True, timeout=600)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN[^\n:]*gemm_(?:tn_)?kernel_pipe[^\n:]*):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) == 4 and sum("kernel_pipe_ax" in k for k, _ in kernels) == 1, [k for k, _ in kernels]

    def vregs(operand_text):
        regs = set()
        for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", operand_text):
            regs.update(range(int(lo), int(hi) + 1))
        regs.update(int(r) for r in re.findall(r"\bv(\d+)\b", operand_text))
        return regs

    for name, body in kernels:
        lines = body.split("\n")
        first_barrier = next(i for i, l in enumerate(lines) if "s_barrier" in l)
        last_drain = max(i for i, l in enumerate(lines) if "s_nop 15" in l)
        # ONE loop behind the prologue barrier; in front of it only the _ax kernel may have one (round 6: the fold of the BatchNorm
        # backward's tile partials into alpha / beta / gamma, before any asm load is issued)
        heads = [i for i, l in enumerate(lines) if "Loop Header" in l]
        assert sum(h > first_barrier for h in heads) == 1 and sum(h < first_barrier for h in heads) == (1 if "kernel_pipe_ax" in name else 0), (name, heads)
        first_asm_load = next(i for i, l in enumerate(lines) if l.strip().startswith("buffer_load"))
        assert all(h < first_asm_load for h in heads if h < first_barrier), (name, heads, first_asm_load)
        head = next(h for h in heads if h > first_barrier)
        label = lines[head].split(":")[0].strip()
        back = next(i for i, l in enumerate(lines) if i > head and "s_cbranch" in l and label in l)
        assert first_barrier < head < back < last_drain
        loop = [l.split(";")[0].strip() for l in lines[head + 1:back]]
        # inside the main loop body: no vector-ALU instruction on vector registers at all
        valu = [l for l in loop if re.match(r"v_(?!mfma)", l) and re.search(r"\b[va]\[?\d", l)]
        ax = "kernel_pipe_ax" in name
        if ax:
            assert len(valu) == 32 and all(l.startswith("v_pk_fma_f32") for l in valu), (name, valu[:5])
        else:
            assert not valu, (name, valu[:5])
        assert sum(l.startswith("v_mfma_f32_32x32x2") for l in loop) == 128      # two k-tiles per iteration
        # prologue barrier .. drain in layout order (the out-of-line block follows the loop body): a register whose LATEST writer is an
        # asm load / LDS read is "pending" -- the compiler cannot know when it lands -- and no vector-ALU instruction may read it
        # (the MFMAs and LDS writes that consume such registers sit behind the hand-placed s_waitcnt).  A VALU write ends the state:
        # the compiler reusing a dead register for address arithmetic in front of the loop is fine.
        # (global loads land in order: an s_waitcnt vmcnt(N) leaves only the N youngest in flight -- what the operand transform of the _ax
        #  kernel relies on; LDS reads are never released here: nothing but MFMAs may consume them)
        pending, clash, side = set(), [], []
        vm_loads = []                                          # register sets of the buffer loads issued so far, oldest first
        for raw in lines[first_barrier + 1:last_drain]:
            l = raw.split(";")[0].strip()
            if not l or l.endswith(":") or l.startswith("."):
                continue
            ops_ = l.split(None, 1)
            args = [a.strip() for a in ops_[1].split(",")] if len(ops_) > 1 else []
            if l.startswith("s_waitcnt") and "vmcnt(" in l:
                keep = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
                landed, vm_loads = (vm_loads[:len(vm_loads) - keep], vm_loads[len(vm_loads) - keep:]) if keep else (vm_loads, [])
                in_flight = set().union(*vm_loads) if vm_loads else set()
                for regs in landed:
                    pending -= regs - in_flight
            elif l.startswith(("buffer_load", "ds_read")):
                pending |= vregs(args[0])
                if l.startswith("buffer_load"):
                    vm_loads.append(vregs(args[0]))
            elif re.match(r"v_(?!mfma)", l):
                side.append(l)
                if any(vregs(a) & pending for a in args[1:]):
                    clash.append(l)
                pending -= vregs(args[0]) if args else set()
        assert side, name
        assert not clash, (name, clash[:5])
        assert sum(l.startswith("v_pk_add_f32") for l in side) == 32, name


def test_wave_walk_gemm_loop_is_the_stated_slot_schedule_and_nothing_else(tmp_path):
    """gemm_rowwalk_kernel (K3w, csrc/gemm_rowpanel.hip, round 5) keeps registers in flight across its whole walk -- A fragments requested a
    tile ahead, W fragments a k-group ahead, accumulators in AGPRs -- all through inline asm the compiler cannot see into.  Checked on the
    ISA hipcc generates for gfx950 (the development subset: K = 100 and K = 128 reductions, with and without ReLU):
      * no spill, no scratch: a spilled register of this kernel would be copied while its load is still in flight;
      * exactly ONE loop, whose body holds the stated schedule and nothing else: KG x 16 MFMAs, 4 KG ds_read_b128, KG buffer loads, 64
        dword stores, 64 accumulator reads -- and outside the asm blocks only scalar bookkeeping and the 64 epilogue FMAs (+ 64 max with
        ReLU): no compiler-placed s_waitcnt (it would be a wait for ALL loads: the software pipeline gone), no scalar / vector memory
        instruction, no register copy;
      * every MFMA of the loop reads its A operand from the rolling fragment registers and accumulates in AGPRs."""
    import re, shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "graphless-neural-networks_amd", "csrc", "gemm_rowpanel.hip")
    out = tmp_path / "rowpanel.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DGLNN_RP_DEV", f"-I{ROOT}/include",
                    f"-I{ROOT}/graphless-neural-networks_amd/csrc", "-S", "--cuda-device-only", "-o", str(out), src],
                   check=True, capture_output=True, timeout=900)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN[^\n:]*gemm_rowwalk_kernelILi(\d+)ELb([01])E[^\n:]*):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert sorted((int(kg), int(relu)) for _, kg, relu, _ in kernels) == [(13, 0), (13, 1), (16, 0), (16, 1)]
    for name, kg, relu, body in kernels:
        kg, relu = int(kg), int(relu)
        meta = text[text.index(f".name:           {name}"):]
        meta = meta[:meta.index(".wavefront_size")]
        assert re.search(r"\.vgpr_spill_count:\s+0\b", meta) and re.search(r"\.sgpr_spill_count:\s+0\b", meta), name
        assert re.search(r"\.private_segment_fixed_size:\s+0\b", meta), name
        assert "scratch_" not in body, name
        lines = body.split("\n")
        # basic blocks in layout order: [label line, next label line); the walk's loop = the header block whose loop holds the MFMAs
        # plus every block the compiler tagged "in Loop: Header=<that block>"
        starts = [i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)] + [len(lines)]
        blocks = [(lines[a0].split(":")[0].lstrip(".L"), lines[a0], lines[a0 + 1:a1]) for a0, a1 in zip(starts[:-1], starts[1:])]
        loops = {}
        for label, first, blk in blocks:
            if "Loop Header" in first:
                loops.setdefault(label, []).extend(blk)
            m_ = re.search(r"in Loop: Header=(BB\d+_\d+)", first)
            if m_:
                loops.setdefault(m_.group(1), []).extend(blk)
        walk = [v for v in loops.values() if any("v_mfma" in l for l in v)]
        assert len(walk) == 1, (name, list(loops))
        loop = walk[0]
        inasm, asm_ops, other = False, [], []
        for raw in loop:
            if "#ASMSTART" in raw:
                inasm = True
                continue
            if "#ASMEND" in raw:
                inasm = False
                continue
            l = raw.split(";")[0].strip()
            if not l or l.endswith(":") or l.startswith("."):
                continue
            (asm_ops if inasm else other).append(l)
        count = lambda ops_, pre: sum(o.startswith(pre) for o in ops_)
        assert count(asm_ops, "v_mfma_f32_32x32x2") == 16 * kg, (name, count(asm_ops, "v_mfma_f32_32x32x2"))
        assert count(asm_ops, "ds_read_b128") == 4 * kg and count(asm_ops, "buffer_load_dwordx4") == kg, name
        assert count(asm_ops, "buffer_store_dword") == 64 and count(asm_ops, "v_accvgpr_read_b32") == 64, name
        # outside the asm blocks: scalar ALU / branches / s_nop, and the epilogue arithmetic -- nothing that waits, loads, stores or copies
        bad = [o for o in other if not re.match(r"(s_(?!waitcnt|load|buffer|sleep|barrier)\w+|v_cmp_\w+|v_fma_f32|v_fmac_f32|v_max_f32\w*)\b", o)]
        assert not bad, (name, bad[:8])
        assert count(other, "v_fma") + count(other, "v_fmac") == 64 and count(other, "v_max_f32") == (64 if relu else 0), name
        for o in asm_ops:
            if o.startswith("v_mfma"):
                a = [x.strip() for x in o.split(None, 1)[1].split(",")]
                assert a[0].startswith("a[") and a[1].startswith("v") and a[2].startswith("v") and (a[3] == "0" or a[3] == a[0]), (name, o)


def test_cross_workgroup_publishes_drain_their_stores_before_the_counter_update(tmp_path):
    """last_workgroup / bn_bwd_fused (csrc/student.hip) publish partial sums with write-through stores and then bump an arrival
    counter that workgroups on OTHER XCDs read.  A barrier alone does not wait for the write-through on gfx950 (the ISA used to be
    `global_store_dword ... sc1; s_barrier; global_atomic_add`), so every storing wave must drain its vector-memory queue first.
    Checked on the generated ISA: in every kernel, walking back from each counter atomic to the last write-through store before it
    crosses an `s_waitcnt vmcnt(0)` (every basic-block path is inspected textually: the store and the waitcnt are emitted in
    straight-line order in front of the barrier)."""
    import shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    text = ""
    for unit in ("student", "mlp_lat"):          # every translation unit that uses the protocol (student_dev.h)
        src = os.path.join(ROOT, "graphless-neural-networks_amd", "csrc", unit + ".hip")
        out = tmp_path / (unit + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", f"-I{ROOT}/include",
                        f"-I{ROOT}/graphless-neural-networks_amd/csrc", "-S", "--cuda-device-only", "-o", str(out), src],
                       check=True, capture_output=True, timeout=600)
        text += out.read_text()
    kernels = re.findall(r"^(_ZN[^\n:]*):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    checked = 0
    for name, body in kernels:
        lines = [l.split(";")[0].strip() for l in body.split("\n")]
        atomics = [i for i, l in enumerate(lines) if l.startswith("global_atomic_add")]
        stores = [i for i, l in enumerate(lines) if l.startswith("global_store") and "sc1" in l]
        if not atomics or not stores:
            continue
        for a in atomics:
            before = [s for s in stores if s < a]
            if not before:
                continue
            region = lines[before[-1] + 1:a]
            assert any(re.match(r"s_waitcnt\s+vmcnt\(0\)", l) for l in region), (name, region[:12])
            assert any(l.startswith("s_barrier") for l in region), name
            checked += 1
    assert checked >= 8, checked      # loss, bn statistics, bn backward (partial / fused), column sums; the latency GEMM's epilogues


def test_library_reads_the_environment_in_one_place():
    """The GLNN_* switches (glnn::Options) are read ONCE, by one function of csrc/capi.hip; no kernel launcher calls getenv
    (round 3 had 14 per-call reads on the hot launch paths)."""
    import glob
    hits = {}
    for f in glob.glob(os.path.join(ROOT, "graphless-neural-networks_amd", "csrc", "*")):
        if f.endswith((".hip", ".h")):
            n = len(re.findall(r"\bgetenv\s*\(", open(f).read()))
            if n:
                hits[os.path.basename(f)] = n
    assert hits == {"capi.hip": 1}, hits


def test_reload_options_follows_the_environment(monkeypatch):
    from glnn_amd import _lib
    h = _lib.lib()
    monkeypatch.setenv("GLNN_GEMM_PIPE", "0")
    h.glnn_reload_options()            # (no way to read a switch back through the C ABI: the GPU tests observe the effect)
    monkeypatch.delenv("GLNN_GEMM_PIPE")
    h.glnn_reload_options()
