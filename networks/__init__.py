"""Drop-in shim: lets the reference's callers keep ``from networks.swinIR_variations import
make_RDSTSR`` (models/trans_sr_trainer.py:3) when this repository is first on ``sys.path``.
Everything is re-exported from ``rdst_amd.networks``."""
