"""The C ABI without Python or torch in the loop: examples/abi_train (HIP runtime + libuwm.so only) owns every device
buffer, builds the model through uwm_create / the tensor table, and trains for a few steps — the loss must fall."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("arch,enc", [(0, 18), (1, 18), (0, 50)])
def test_torch_free_host_program_trains(cuda, arch, enc):
    import __graft_entry__ as g
    exe = g.build_abi_example()
    r = subprocess.run([exe, str(arch), str(enc), "4", "128", "128", "5"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, GPU_MAX_HW_QUEUES="8"))
    assert r.returncode == 0, r.stdout + r.stderr
    losses = [float(m) for m in re.findall(r"dice_loss ([0-9.]+)", r.stdout)]
    assert len(losses) == 5 and all(l == l and 0.0 <= l <= 1.0 for l in losses), r.stdout
    assert losses[-1] < losses[0] - 0.02, losses
    # the binary links the HIP runtime and libuwm only
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libuwm.so" in ldd and "libtorch" not in ldd and "libc10" not in ldd
