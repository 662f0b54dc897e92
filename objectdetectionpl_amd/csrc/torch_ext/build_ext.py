"""Builds odpl_torch_ext*.so in-tree (g++, no device code: it only forwards to libodpl_hip.so)."""
import os
import sys


def build(verbose=False):
    import torch  # noqa: F401
    from torch.utils import cpp_extension
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.dirname(here)
    os.environ.setdefault("MAX_JOBS", "4")
    cpp_extension.load(
        name="odpl_torch_ext", sources=[os.path.join(here, "odpl_torch.cpp")], build_directory=here, verbose=verbose,
        extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"], extra_include_paths=["/opt/rocm/include"],
        extra_ldflags=[f"-L{csrc}", "-lodpl_hip", f"-Wl,-rpath,{csrc}", "-L/opt/rocm/lib", "-lamdhip64"],
        with_cuda=False, is_python_module=True)
    return os.path.join(here, "odpl_torch_ext.so")


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
