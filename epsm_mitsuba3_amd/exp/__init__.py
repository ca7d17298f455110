"""Experiment configurations in the shape of EPSM/exp/*.py (module-level constants + ``optim_settings()``).
The reference's scenes need mesh / texture assets that are not in its repository (README.md:26); the
configurations here are analytic stand-ins built from inline meshes."""
