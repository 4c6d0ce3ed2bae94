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
