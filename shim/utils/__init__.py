"""`utils` package of the reference, hot-path modules only (sampler, common), backed by edtr_amd."""
