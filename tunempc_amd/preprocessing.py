"""Boundary validation of the convexifier inputs (reference: tunempc/preprocessing.py:157-185)."""


def input_checks(arg):
    """Input checks for provided convexification matrices A, B, Q, R, N, G, (C).

    - all provided matrices are of the same type (list vs. single matrix)
    - sizes are consistent along the trajectory (C exempt: ragged active sets)
    - matrices are returned as lists
    Same assertion messages as the reference (preprocessing.py:167,171,178).
    """
    msg1 = "Input arguments should be of same type!"
    assert (all(type(argument) == type(arg['A']) for key, argument in arg.items())), msg1

    if type(arg['A']) == list:
        msg2 = "Input data lists should have same length!"
        assert (all(len(argument) == len(arg['A']) for key, argument in arg.items())), msg2
    else:
        for key in list(arg.keys()):
            arg[key] = [arg[key]]

    msg3 = "Data matrices should have same size along trajectory."
    for key, argument in arg.items():
        if key != 'C':
            assert (all(_shape(mat) == _shape(argument[0]) for mat in argument)), msg3
    return arg


def _shape(m):
    return tuple(getattr(m, 'shape', ()))
