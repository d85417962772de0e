from .interp_fn import cubic_hermite_interp, linear_interp  # noqa: F401
