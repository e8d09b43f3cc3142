"""Minimal stand-in for ``numexpr`` (absent from the system interpreter): the reference imports it unconditionally in
``pymbar/utils.py`` although the solver path never uses it.  ``evaluate`` runs the expression with numpy in the caller's
frame, which is all the reference's ``utils.logsumexp`` needs."""
import sys

import numpy as np

__version__ = "2.8.4"  # (pandas parses the version of an optional numexpr)


def evaluate(expr, local_dict=None, global_dict=None, **kwargs):
    frame = sys._getframe(1)
    env = dict(np.__dict__)
    env.update(frame.f_globals if global_dict is None else global_dict)
    env.update(frame.f_locals if local_dict is None else local_dict)
    return eval(expr, env)
