"""Build-time invariants of the HIP sources that the GPU tests cannot see.

The matrix-core GEMV waits on its weight stream with hand-counted `s_waitcnt vmcnt(N)`.  A
register spill (or a dynamically indexed private array) would become scratch traffic = extra VMEM
operations the counts do not know about, i.e. silent use of not-yet-loaded registers.  So every
instantiation must compile without scratch."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_gemv_kernels_use_no_scratch_and_touch_no_register_in_flight():
    """e8p_gemv_mfma.hip, every instantiation: no scratch, and (tools/check_inflight.py) no instruction reads or writes
    a register that a counted load is still going to write -- the slots are tied asm operands here, so a copy ahead
    of the wait is exactly what the allocator could emit; the E8P12RVQ3B modes stream 12-byte slots"""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_inflight
    src = os.path.join(REPO, "quip_for_all_amd", "csrc", "e8p_gemv_mfma.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-", src,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, check=True)
    name, seen, bad = None, 0, []
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and "e8p_gemv_mfma_kernel" in name:
            seen += 1
            if int(m.group(1)) != 0:
                bad.append((name, int(m.group(1))))
    assert seen >= 20, "resource remarks not found"
    assert not bad, bad
    kernels = [(n, l) for n, l in check_inflight.kernels_of(r.stdout) if "e8p_gemv_mfma_kernel" in n]
    assert len(kernels) >= 20
    native = 0
    for kname, lines in kernels:
        assert check_inflight.check_kernel(lines) == [], kname
        rvq3 = re.search(r"e8p_gemv_mfma_kernelILi(40|20)E", kname) is not None
        has3 = any("global_load_dwordx3" in l for l in lines)
        assert rvq3 == has3, kname          # the RVQ3 modes, and only they, load 12-byte pieces
        native += rvq3
    assert native >= 10


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_gemv_v2_kernels_use_no_scratch_and_only_counted_loads():
    """e8p_gemv_v2.hip: no scratch, and no compiler-generated VMEM load ahead of the stream (a plain load of an
    invariant kernel-argument word was once hoisted into the prologue, where it shifted every counted wait by one):
    between the kernel entry and the last `global_load_dwordx4 ... nt` all loads are the asm ones."""
    src = os.path.join(REPO, "quip_for_all_amd", "csrc", "e8p_gemv_v2.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-", src,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, check=True)
    name, seen, bad = None, 0, []
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and "e8p_gemv_v2_kernel" in name:
            seen += 1
            if int(m.group(1)) != 0:
                bad.append((name, int(m.group(1))))
    assert seen >= 30, "resource remarks not found"
    assert not bad, bad
    kernels = re.findall(r"^(_ZN4quip\S*e8p_gemv_v2_kernel\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel", r.stdout, re.S | re.M)
    assert len(kernels) >= 30
    for kname, body in kernels:
        m = re.match(r".*e8p_gemv_v2_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", kname)
        slots, g = int(m.group(3)), int(m.group(4))
        rvq3 = "Lb1E" in kname      # the third-table mode: 12-byte slots (dwordx3), a second table source load
        lines = body.splitlines()
        wide = "global_load_dwordx3" if rvq3 else "global_load_dwordx4"
        last_nt = max(i for i, l in enumerate(lines) if wide in l and " nt" in l)
        loads = [l.strip() for l in lines[:last_nt + 1] if re.search(r"\b(global|buffer|scratch|flat)_load", l)]
        # prologue: G shift words, 1 table entry (2 with the third table), up to 6 + 6 digit pieces (filler / real
        # branch), SLOTS weight loads; stream: SLOTS reloads
        assert sum("global_load_dword " in l for l in loads) == g, (kname, loads)
        assert sum("global_load_dwordx2" in l for l in loads) == (2 if rvq3 else 1), kname
        assert sum(" nt" in l for l in loads) == 2 * slots, kname
        assert all((wide in l) for l in loads if " nt" in l), kname
        assert all(l.startswith(("global_load_dword ", "global_load_dwordx2", "global_load_dwordx3", "global_load_dwordx4"))
                   for l in loads)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_gemv_v2_nibble_kernels_use_no_scratch_and_only_counted_loads():
    """e8p_gemv_v2n.hip (round 6: the K-splitting kernel in nibble mode), all nine instantiations (2 / 3 / 4 slots x 1 / 2 / 3
    problems): no scratch, and every load up to the last weight request is one of the counted asm loads -- G shift words, one
    table entry pair, 3 G NG digit pieces off scalar plane bases (NG = 2 for one problem, else 1), SLOTS requests + SLOTS reloads."""
    src = os.path.join(REPO, "quip_for_all_amd", "csrc", "e8p_gemv_v2n.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-", src,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, check=True)
    name, seen, bad = None, 0, []
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and "e8p_gemv_v2n_kernel" in name:
            seen += 1
            if int(m.group(1)) != 0:
                bad.append((name, int(m.group(1))))
    assert seen == 9, seen
    assert not bad, bad
    kernels = re.findall(r"^(_ZN4quip\S*e8p_gemv_v2n_kernel\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel", r.stdout, re.S | re.M)
    assert len(kernels) == 9
    for kname, body in kernels:
        m = re.match(r".*e8p_gemv_v2n_kernelILi(\d+)ELi(\d+)E", kname)
        slots, g = int(m.group(1)), int(m.group(2))
        lines = body.splitlines()
        last_nt = max(i for i, l in enumerate(lines) if "global_load_dwordx4" in l and " nt" in l)
        loads = [l.strip() for l in lines[:last_nt + 1] if re.search(r"\b(global|buffer|scratch|flat)_load", l)]
        assert sum("global_load_dword " in l for l in loads) == g, (kname, loads)
        assert sum("global_load_dwordx2" in l for l in loads) == 1, kname
        assert sum(" nt" in l for l in loads) == 2 * slots, kname
        ng = 2 if g == 1 else 1
        pieces = [l for l in loads if "global_load_dwordx4" in l and " nt" not in l]
        assert len(pieces) == 3 * g * ng, (kname, len(pieces))
        assert all(re.search(r", s\[\d+:\d+\]", l) for l in pieces), "digit pieces off scalar plane bases"
        assert all(l.startswith(("global_load_dword ", "global_load_dwordx2", "global_load_dwordx4")) for l in loads)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_skinny_kernel_touches_no_register_in_flight():
    """e8p_skinny_gemm.hip counts its vector-memory queue by hand around asm loads, so the compiler does not know
    which registers are still going to be written.  A first version tied such registers to the wait ("+v"): in some
    instantiations the allocator then copied them BEFORE the wait, and one launch in a few hundred multiplied stale
    codes.  tools/check_inflight.py follows every path of the compiled kernels with the in-order vmcnt queue and
    reports any instruction that reads or writes the destination of a load still in it; also: no scratch."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_inflight
    src = os.path.join(REPO, "quip_for_all_amd", "csrc", "e8p_skinny_gemm.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-", src,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, check=True)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert len(scratch) == 20 and not any(scratch), scratch     # 2 x 2 shapes x five codebook modes
    kernels = [(n, l) for n, l in check_inflight.kernels_of(r.stdout) if "e8p_skinny_gemm_kernel" in n]
    assert len(kernels) == 20
    for name, lines in kernels:
        assert any("global_load_lds_dwordx4" in l for l in lines), name
        assert check_inflight.check_kernel(lines) == [], name


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_prefill_tile_kernels_of_every_codebook_use_no_scratch():
    """e8p_prefill_gemm.hip: the eight-wave layout (the one the launcher selects) in all five codebook modes keeps its 64 / 128
    accumulator registers, the code registers in flight and the table entries in VGPRs -- a spill would put scratch
    (VMEM) operations into a K loop whose vmcnt queue is counted by hand.  (tools/check_inflight.py is not run on this
    file: its `if (t + i < KT) tile(..)` loop has control-flow paths from a skipped tile back to the loop head that
    never execute -- t + i >= KT ends the loop -- and the checker follows every path.)"""
    src = os.path.join(REPO, "quip_for_all_amd", "csrc", "e8p_prefill_gemm.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", os.devnull, src,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, check=True)
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert len(names) == len(scratch)
    eight_wave = {n: sc for n, sc in zip(names, scratch) if "e8p_prefill_gemm_kernel" in n and "ELi1ELi1ELi8ELi" in n}
    # <4 | 8 row blocks> x E8P12, D4, HI; 128-row tiles only for the two RVQ codebooks
    assert len(eight_wave) == 8, sorted(eight_wave)
    assert not any(eight_wave.values()), eight_wave


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_block_engine_kernels_use_no_scratch_and_touch_no_register_in_flight():
    """decode_block.hip keeps weight requests in flight in asm-written registers across whole phases of the persistent
    launch, and sits within a few registers of the 256 a 512-thread workgroup can have: a spill costs 10-30 us per block
    (measured), and a register copied while its load is in flight is a wrong token once in a while.  All seven
    instantiations (E8P12 in nibble mode -- round 6, the shipped one -- and with 24 / 16 byte-table copies, D4, E8P12RVQ4B, HI,
    E8P12RVQ3B): no scratch, no instruction on an in-flight register."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_inflight
    src = os.path.join(REPO, "quip_for_all_amd", "csrc", "decode_block.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-", src,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, check=True)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert len(scratch) == 7 and not any(scratch), scratch
    kernels = [(n, l) for n, l in check_inflight.kernels_of(r.stdout) if "decode_block_kernel" in n]
    assert len(kernels) == 7
    for name, lines in kernels:
        assert check_inflight.check_kernel(lines) == [], name


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_g8_block_engine_kernel_uses_no_scratch_and_touches_no_register_in_flight():
    """decode_block_g8.hip: the same launch compiled for the 4096-wide grouped-query shape (Llama-3-8B / Mistral-7B; seven gate / up
    items in flight at once): no scratch, no instruction on an in-flight register"""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_inflight
    src = os.path.join(REPO, "quip_for_all_amd", "csrc", "decode_block_g8.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-", src,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, check=True, cwd=os.path.dirname(src))
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert len(scratch) == 2 and not any(scratch), scratch          # (nibble mode + the byte tables kept for A/B)
    kernels = [(n, l) for n, l in check_inflight.kernels_of(r.stdout) if "decode_block_kernel" in n]
    assert len(kernels) == 2
    for name, lines in kernels:
        assert check_inflight.check_kernel(lines) == [], name


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_gqa_block_engine_kernel_uses_no_scratch_and_touches_no_register_in_flight():
    """decode_block_gqa.hip keeps a ring of nine weight requests per wave in flight through the whole launch and waits with
    the constant `s_waitcnt vmcnt(16)`: no scratch (a spill is a VMEM operation the count does not know), no instruction on a
    register a counted load is still going to write (tools/check_inflight.py follows every path with the in-order queue and
    retires entries at each counted wait), and every ring request is the asm `global_load_dwordx4 ... nt` pair"""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_inflight
    src = os.path.join(REPO, "quip_for_all_amd", "csrc", "decode_block_gqa.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-", src,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, check=True)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert len(scratch) == 1 and scratch[0] == 0, scratch
    kernels = [(n, l) for n, l in check_inflight.kernels_of(r.stdout) if "decode_block_gqa_kernel" in n]
    assert len(kernels) == 1
    name, lines = kernels[0]
    assert check_inflight.check_kernel(lines) == [], name
    waits = [l for l in lines if l.startswith("s_waitcnt") and "vmcnt(16)" in l]
    # EXACTLY one per item of the sequence (the block loop's body is one iteration): a product that forgets the wait / refill of
    # one of its items (round 6: a filler dropped while the products were re-ordered) shifts every later wait by one slot
    # (round 6: down's second item decoded ahead waits with vmcnt(14) -- no refill in front of it, see predec())
    w14 = [l for l in lines if l.startswith("s_waitcnt") and "vmcnt(14)" in l]
    assert len(waits) == 53 and len(w14) >= 1, (len(waits), len(w14))
    nt = [l for l in lines if "global_load_dwordx4" in l and " nt" in l]
    assert len(nt) >= 2 * (54 + 9) and len(nt) % 2 == 0


def test_inflight_checker_flags_a_copy_before_the_wait():
    """the checker itself, on a hand-written listing: a tied-operand copy ahead of the wait is reported, the same
    sequence with the wait first is clean, and a conditional skip of the wait is followed"""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_inflight
    bad = ["global_load_dwordx4 v[22:25], v[36:37], off", "global_load_lds_dwordx4 v[2:3], off", "v_mov_b32_e32 v26, v22",
           "s_waitcnt vmcnt(1)", "v_add_u32_e32 v1, v26, v23", "s_endpgm"]
    got = check_inflight.check_kernel(bad)
    assert [g[0] for g in got] == [2]
    good = [bad[0], bad[1], bad[3], bad[2], bad[4], bad[5]]
    assert check_inflight.check_kernel(good) == []
    skip = ["global_load_dwordx2 v[4:5], v[0:1], off", "s_cbranch_scc1 .L1", "s_waitcnt vmcnt(0)", ".L1:",
            "v_add_u32_e32 v6, v4, v5", "s_endpgm"]
    assert [g[0] for g in check_inflight.check_kernel(skip)] == [4]
