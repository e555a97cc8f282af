"""In-tree build of the HIP extension (and, for the checker, nothing else)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=False, jobs=4):
    """Compile categoricalnf_amd/csrc/*.hip for gfx950 into categoricalnf_amd/lib/libcnf_hip.so."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j%d" % jobs]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libcnf_hip.so failed (exit %d)" % res.returncode)
    return os.path.join(_HERE, "lib", "libcnf_hip.so")
