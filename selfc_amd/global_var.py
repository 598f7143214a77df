"""Process-global clip length, same contract as the reference's codes/global_var.py:3-30.

The reference's subnets read the temporal length from this class at forward time
(Subnet_constructor.py:121, SelfC_GMM_arch_inv.py:376,460,472) and the dataset
constructor sets it (data/LQGTVID_dataset.py:50).  When the reference's own
``global_var`` module is importable (drop-in use inside its tree) its class is
re-exported so both sides see one value.
"""
import random
import string
import sys


def _accessor(attr):
    """(getter, setter) pair over a class attribute of whichever GlobalVar class is exported below; an attribute that was
    never set reads as None, as in the reference."""

    def getter():
        return GlobalVar.__dict__.get(attr)

    def setter(value):
        setattr(GlobalVar, attr, value)

    return staticmethod(getter), staticmethod(setter)


class _GlobalVar:
    def __init__(self):      # the reference declares an empty constructor (global_var.py:4-5); nothing is ever instantiated
        pass

    get_Temporal_LEN, set_Temporal_LEN = _accessor("VIDEO_T_LEN")      # frames per clip (7 for septuplets)
    get_Istrain, set_Istrain = _accessor("Istrain")

    @staticmethod
    def get_v_random_name():
        """Five distinct letters, drawn once per process (temp-file tag of the codec variant)."""
        name = GlobalVar.__dict__.get("encode_video_random_name")
        if name is None:
            name = "".join(random.sample(string.ascii_lowercase[::-1], 5))
            GlobalVar.encode_video_random_name = name
        return name


_ref = sys.modules.get("global_var")
GlobalVar = _ref.GlobalVar if _ref is not None and hasattr(_ref, "GlobalVar") else _GlobalVar
