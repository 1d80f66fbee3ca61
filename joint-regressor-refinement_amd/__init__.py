"""MI355X-native pose-refinement hot path of ubc-vision/joint-regressor-refinement.

Import as  importlib.import_module("joint-regressor-refinement_amd")  (the directory name is fixed
by the project layout) or through the `jrr_amd` alias module at the repository root.

Layout:
  csrc/            HIP kernels + C ABI (include/jrr.h) -> libjrr_hip.so (build.py)
  _lib.py          ctypes binding (no CPU fallback: raises if the library is missing)
  engine.py        DeviceModel / RefineEngine wrappers over the C ABI
  smpl_model.py    SMPL constants: real-model loader + seeded synthetic generator
  smpl.py utils.py discriminator.py optimize.py args.py
                   host-side mirror of the reference's scripts/ modules for this path
"""
__version__ = '0.1.0'

from . import smpl_model  # noqa: F401  (numpy only)


def __getattr__(name):
    import importlib
    if name in ('engine', '_lib', 'build', 'smpl', 'utils', 'discriminator', 'optimize', 'args', 'checkpoint', 'dist', 'test', 'data', 'renderer',
                'mesh_renderer'):
        return importlib.import_module(f'{__name__}.{name}')
    raise AttributeError(name)
