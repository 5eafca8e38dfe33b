"""In-tree build of libtunempc_hip.so for gfx950 (explicit hipcc, no torch extension machinery:
the library has a plain C ABI and links only the HIP runtime)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'tmpc_api.hip')
OUT = os.path.join(HERE, 'lib', 'libtunempc_hip.so')


def _newer(out, deps):
    if not os.path.exists(out):
        return False
    t = os.path.getmtime(out)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False):
    deps = [os.path.join(HERE, 'csrc', f) for f in os.listdir(os.path.join(HERE, 'csrc'))]
    deps.append(os.path.join(os.path.dirname(HERE), 'include', 'tunempc_hip.h'))
    if not force and _newer(OUT, deps):
        return OUT
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        if os.path.exists(OUT):
            return OUT            # prebuilt library shipped with the snapshot (GPU box without a toolchain)
        raise RuntimeError('hipcc not found and no prebuilt libtunempc_hip.so')
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-Wno-unused-value', SRC, '-o', OUT]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force=True, verbose=True))
