"""pip install -e .  -- builds the gfx950 libraries in-tree (make -C bokego_amd/csrc) and installs the Python
host package.  The reference installs the same way (its setup.py lists torch, numpy, pandas); here only numpy and
torch (ROCm build) are needed at run time."""
import os
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class BuildWithMake(build_py):
    def run(self):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "bokego_amd", "csrc")])
        super().run()


setup(
    name="bokego_amd",
    version="0.1.0",
    description="MI355X-native leaf-evaluation engine for bokego's MCTS (HIP kernels behind a C ABI)",
    packages=["bokego_amd"],
    package_data={"bokego_amd": ["*.so"]},
    python_requires=">=3.10",
    install_requires=["numpy"],
    cmdclass={"build_py": BuildWithMake},
)
