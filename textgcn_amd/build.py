"""Build recipe of libtgcn.so (HIP kernels + C ABI), in-tree, gfx950 only.

    python -m textgcn_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU, so the build container produces the .so that travels to the GPU box
(`*.so` is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, 'csrc')
LIB_DIR = os.path.join(_HERE, '_lib')
LIB = os.path.join(LIB_DIR, 'libtgcn.so')
SOURCES = ['tgcn_core.hip', 'tgcn_spmm.hip', 'tgcn_score.hip', 'tgcn_score_fused.hip', 'tgcn_score_prefilter.hip', 'tgcn_ltr.hip', 'tgcn_comm.hip', 'tgcn_train.hip']
HEADERS = [os.path.join(ROOT, 'include', 'tgcn.h'), os.path.join(CSRC, 'tgcn_internal.h'), os.path.join(CSRC, 'tgcn_topk.h')]
ARCH = 'gfx950'


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def torch_lib_dir():
    """PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, SONAME without version).
    libtgcn.so must bind to THAT copy: kernels are launched on torch's streams with torch's allocations, and
    a second runtime instance in the process (/opt/rocm's libamdhip64.so.7) sees no device."""
    import importlib.util
    spec = importlib.util.find_spec('torch')
    return os.path.join(os.path.dirname(spec.origin), 'lib')


def compile_flags():
    return [f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC',
            '-ffp-contract=off',  # fused multiply-adds are written explicitly (fmaf / MFMA); nothing else may fuse
            '-fno-fast-math', f'-I{os.path.join(ROOT, "include")}', f'-I{CSRC}']


def link_flags():
    tl = torch_lib_dir()
    return ['-shared', '-fPIC', '--hip-link', f'--offload-arch={ARCH}', '-no-hip-rt', f'-L{tl}', '-lamdhip64', '-ldl',
            f'-Wl,-rpath,{tl}', '-Wl,-rpath,/opt/rocm/lib']


def _obj(src):
    return os.path.join(LIB_DIR, 'obj', os.path.splitext(src)[0] + '.o')


def _obj_stale(src):
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in [os.path.join(CSRC, src)] + HEADERS + [os.path.abspath(__file__)])


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    """One object per translation unit (compiled in parallel, rebuilt only when its source or a header changed), then
    one link: an edit to one kernel file costs that file's compile time, not the whole library's."""
    if not force and not stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(os.path.join(LIB_DIR, 'obj'), exist_ok=True)
    cc = hipcc()

    def compile_one(src):
        cmd = [cc] + compile_flags() + ['-c', os.path.join(CSRC, src), '-o', _obj(src)]
        if verbose:
            cmd.append('-Rpass-analysis=kernel-resource-usage')
            print(' '.join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    todo = [s for s in SOURCES if force or _obj_stale(s)]
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(todo)))) as pool:
        list(pool.map(compile_one, todo))
    cmd = [cc] + [_obj(s) for s in SOURCES] + link_flags() + ['-o', LIB]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build_lib(force='--force' in sys.argv, verbose='--verbose' in sys.argv))
