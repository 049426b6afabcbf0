"""Packaging + build hook: the shared library is a Makefile target (hipcc, gfx950), built before the files are collected."""
import os
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class BuildWithHipLibrary(build_py):
    def run(self):
        subprocess.check_call(["make", "-C", HERE])  # -> polars_strsim/libpolars_strsim_amd.so (package data)
        super().run()


setup(
    name="polars-strsim-amd",
    version="0.2.3",  # the reference release whose expression surface this mirrors (pyproject.toml:6)
    description="polars-strsim's five string-similarity expressions on AMD MI355X (gfx950): plugin library + the same wrappers",
    python_requires=">=3.8",
    install_requires=["polars>=1.0,<2.0"],
    packages=["polars_strsim", "strsim_amd"],
    package_data={"polars_strsim": ["libpolars_strsim_amd.so"]},
    cmdclass={"build_py": BuildWithHipLibrary},
    zip_safe=False,
)
