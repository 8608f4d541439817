


# Fall-through to the reference checkout: this package shadows the reference's directory of the same name (it must come first on
# sys.path so that the hot-path modules resolve here), but train.py / test.py also import sibling modules that are NOT on the hot
# path (utils.io, utils.training, utils.evaluate, diffusion.resample, models.modules ...).  Appending the same-named directories
# found further down sys.path to __path__ lets those resolve to the reference's files, while modules defined here win.
def _extend_path_with_reference():
    import os
    import sys
    here = os.path.abspath(os.path.dirname(__file__))
    name = __name__.split(".")[-1]
    for entry in list(sys.path):
        cand = os.path.abspath(os.path.join(entry or ".", name))
        if cand != here and os.path.isdir(cand) and cand not in __path__:
            __path__.append(cand)


_extend_path_with_reference()
