"""Solver <-> equation attribute plumbing (the reference's pde_opt/utils.py:6-53)."""


def check_equation_solver_compatibility(solver_type, equation_type):
    """``ValueError`` when the equation class lacks an attribute the solver will ask for."""
    wanted = getattr(solver_type, "required_equation_attrs", None)
    if not wanted:
        return
    missing = [name for name in wanted if not hasattr(equation_type, name)]
    if missing:
        raise ValueError(
            f"Equation type {equation_type.__name__} is missing required "
            f"attributes for solver {solver_type.__name__}: {missing}"
        )


def prepare_solver_params(solver_type, solver_parameters, equation):
    """Solver kwargs = user kwargs + the attributes the solver pulls off the equation."""
    merged = dict(solver_parameters)
    for name in getattr(solver_type, "required_equation_attrs", None) or ():
        merged[name] = getattr(equation, name)
    return merged
