"""Boundary validation of the convexifier inputs (behaviour of reference tunempc/preprocessing.py:157-185)."""

MSG_TYPE = "Input arguments should be of same type!"                       # preprocessing.py:167
MSG_LENGTH = "Input data lists should have same length!"                   # preprocessing.py:171
MSG_SHAPE = "Data matrices should have same size along trajectory."        # preprocessing.py:178


def _shape(m):
    return tuple(getattr(m, 'shape', ()))


def input_checks(arg):
    """Validate the dict of convexification inputs (keys A, B, Q, R, N and optionally G, C) and normalise it.

    Either every entry is a list with one matrix per stage, or every entry is a bare matrix (period 1); mixing the two is
    an AssertionError, as are lists of different lengths and, for every key but 'C', matrices whose shape changes along the
    trajectory ('C' holds the ragged active sets, with None for a stage without any).  Bare matrices come back wrapped in
    one-element lists; the dict is modified in place and returned.  The three assertion messages are the reference's."""
    kind = type(arg['A'])
    for value in arg.values():
        assert type(value) == kind, MSG_TYPE
    if kind is list:
        period = len(arg['A'])
        for value in arg.values():
            assert len(value) == period, MSG_LENGTH
    else:
        for key in tuple(arg):
            arg[key] = [arg[key]]
    for key, stages in arg.items():
        if key == 'C':
            continue
        first = _shape(stages[0])
        for mat in stages:
            assert _shape(mat) == first, MSG_SHAPE
    return arg
