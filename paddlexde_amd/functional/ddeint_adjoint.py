"""Adjoint of the delay-equation caller.

The reference declares the name and stops there (paddlexde/functional/ddeint_adjoint.py:1-2 raises
NotImplementedError unconditionally; example/dde_demo.py only reaches it behind ``--adjoint``).  The name is kept so that
``from paddlexde_amd.functional import ddeint_adjoint`` works, and the call fails the same way, with a pointer to what does
work: back-propagating through ``ddeint`` with a fixed-step solver (discretise-then-optimise), which reaches both the
parameters of ``func`` and the lags.
"""

_MESSAGE = (
    "ddeint_adjoint is not implemented (nor is it in the reference); differentiate through "
    "ddeint(..., solver=<fixed-step solver>) instead"
)


def ddeint_adjoint(*args, **kwargs):
    raise NotImplementedError(_MESSAGE)
