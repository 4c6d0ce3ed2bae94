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
def test_gemv_kernels_use_no_scratch():
    src = os.path.join(REPO, "quip_for_all_amd", "csrc", "e8p_gemv_mfma.hip")
    err = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-o", os.devnull, src,
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, check=True).stderr
    name, seen, bad = None, 0, []
    for line in err.splitlines():
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
        lines = body.splitlines()
        last_nt = max(i for i, l in enumerate(lines) if "global_load_dwordx4" in l and " nt" in l)
        loads = [l.strip() for l in lines[:last_nt + 1] if re.search(r"\b(global|buffer|scratch|flat)_load", l)]
        # prologue: G shift words, 1 table entry, up to 6 + 6 digit pieces (filler / real branch), SLOTS weight
        # loads; stream: SLOTS reloads
        assert sum("global_load_dword " in l for l in loads) == g, (kname, loads)
        assert sum("global_load_dwordx2" in l for l in loads) == 1, kname
        assert sum(" nt" in l for l in loads) == 2 * slots, kname
        assert all(l.startswith(("global_load_dword ", "global_load_dwordx2", "global_load_dwordx4")) for l in loads)
