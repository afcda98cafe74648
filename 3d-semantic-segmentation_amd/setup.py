"""Builds the compiled extension module ``project_features_cuda`` for MI355X (gfx950).

Counterpart of the reference's setup.py (cuda_project_image_to_sparse_voxel/setup.py:10-27: a CUDAExtension named
'project_features_cuda' from project_image_cuda.cpp + project_image_cuda_kernel.cu).  Same module name, same entry
point, same commands:

    python setup.py build_ext --inplace      # project_features_cuda.*.so + libvoxproj.so next to this file
    python setup.py install                  # both into site-packages

Two artefacts instead of one, because the kernels sit behind a C-ABI (include/voxproj.h) that non-Python callers use too:

  libvoxproj.so            csrc/voxproj.hip + csrc/vp_*.h, hipcc --offload-arch=gfx950 (csrc/Makefile; the parity contract's
                           -ffp-contract=off and correctly rounded divide/sqrt flags live there)
  project_features_cuda.so csrc/project_features_ext.cpp -- the pybind11 wrapper (argument checks of
                           project_image_cuda.cpp:38-61, current HIP stream, one C-ABI call), linked to libvoxproj.so

The sources are written for HIP directly: nothing is hipified, so the wrapper is a plain CppExtension and the device code
is compiled by the Makefile, not by torch's CUDAExtension machinery (which would run hipify over the tree on ROCm).
"""
import os
import shutil
import subprocess

import torch
from setuptools import setup
from torch.utils.cpp_extension import BuildExtension, CppExtension, include_paths

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")

# Detect PyTorch C++ ABI (setup.py:6-8 of the reference)
cxx11_abi = int(torch.compiled_with_cxx11_abi())
abi_flag = f"-D_GLIBCXX_USE_CXX11_ABI={cxx11_abi}"
print(f"[setup.py] Using ABI flag: {abi_flag}")


class BuildWithHipKernels(BuildExtension):
    def run(self):
        subprocess.check_call(["make", "-C", CSRC, "-s"])           # -> HERE/libvoxproj.so
        super().run()
        if not self.inplace:                                        # install / bdist: ship the kernels beside the module
            os.makedirs(self.build_lib, exist_ok=True)
            shutil.copy2(os.path.join(HERE, "libvoxproj.so"), os.path.join(self.build_lib, "libvoxproj.so"))


setup(
    name="project_features_cuda",
    version="1.0",
    ext_modules=[
        CppExtension(
            "project_features_cuda",
            [os.path.join("csrc", "project_features_ext.cpp")],
            include_dirs=[os.path.join(os.path.dirname(HERE), "include")] + include_paths("cuda"),
            define_macros=[("USE_ROCM", "1"), ("__HIP_PLATFORM_AMD__", "1")],
            library_dirs=[HERE],
            libraries=["voxproj", "c10_hip", "torch_hip"],
            extra_compile_args=["-O3", "-std=c++17", "-w", abi_flag],
            extra_link_args=["-Wl,-rpath,$ORIGIN"],
        ),
    ],
    cmdclass={"build_ext": BuildWithHipKernels},
)
