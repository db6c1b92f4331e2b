"""Drop-in counterparts of the reference's ``networks`` package for the RDST hot path:
``swin_transformer_sr`` (Swin primitives), ``common`` (conv / upsampler / mean shift),
``rdst_variations`` and ``swinIR_variations`` (the module path the reference's trainer imports)."""
