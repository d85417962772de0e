"""The reference's import smoke test (tests/test_import.py:4-8: `import paddlexde; assert hasattr(paddlexde, "__version__")`) and the names
a user script imports — exactly those the reference's package files export for this path (paddlexde/functional/__init__.py:1-4,
solver/__init__.py:1-6, solver/fixed_solver/__init__.py:1-4, solver/adaptive_solver/__init__.py:1-5, xde/__init__.py:2-5); the SDE / CDE
entry points and the SciPy wrapper are out of scope and absent."""
import importlib


def test_import():
    import paddlexde_amd

    assert hasattr(paddlexde_amd, "__version__")


def test_exported_names_are_the_references():
    want = {
        "paddlexde_amd.functional": ["ddeint", "ddeint_adjoint", "odeint", "odeint_adjoint"],
        "paddlexde_amd.solver": ["AdaptiveHeun", "Bosh3", "Dopri5", "Dopri8", "Fehlberg2", "AdaptiveSolver", "AdaptiveRKSolver", "FixedSolver",
                                 "RK4", "AdamsBashforthMoulton", "Euler", "Midpoint"],
        "paddlexde_amd.solver.fixed_solver": ["AdamsBashforthMoulton", "Euler", "Midpoint", "RK4"],
        "paddlexde_amd.solver.adaptive_solver": ["AdaptiveHeun", "Bosh3", "Dopri5", "Dopri8", "Fehlberg2"],
        "paddlexde_amd.xde": ["BaseDDE", "BaseODE", "BaseXDE"],
        "paddlexde_amd.interpolation": ["BezierSpline", "CubicHermiteSpline", "LinearInterpolation"],  # interpolation/__init__.py:1
    }
    for mod, names in want.items():
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), (mod, n)
    top = importlib.import_module("paddlexde_amd")  # (`from .functional import *`, `from .solver import *`, `from .xde import *`: paddlexde/__init__.py:4-8)
    for n in want["paddlexde_amd.functional"] + want["paddlexde_amd.solver"] + want["paddlexde_amd.xde"] + want["paddlexde_amd.interpolation"]:
        assert hasattr(top, n), n
    for n in ("sdeint", "sdeint_adjoint", "ScipyWrapperODESolver", "BaseSDE", "BaseCDE"):  # out of scope: absent, not stubbed
        assert not hasattr(top, n), n
