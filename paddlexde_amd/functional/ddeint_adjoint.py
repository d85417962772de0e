def ddeint_adjoint(**kwargs):
    """Reference: paddlexde/functional/ddeint_adjoint.py:1-2 — not implemented there either."""
    raise NotImplementedError
