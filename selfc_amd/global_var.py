"""Process-global clip length, same contract as the reference's codes/global_var.py:3-30.

The reference's subnets read the temporal length from this class at forward time
(Subnet_constructor.py:121, SelfC_GMM_arch_inv.py:376,460,472) and the dataset
constructor sets it (data/LQGTVID_dataset.py:50).  When the reference's own
``global_var`` module is importable (drop-in use inside its tree) its class is
re-exported so both sides see one value.
"""
import random
import sys


class _GlobalVar:
    VIDEO_T_LEN = None
    Istrain = None

    @staticmethod
    def get_Temporal_LEN():
        return getattr(GlobalVar, "VIDEO_T_LEN", None)

    @staticmethod
    def set_Temporal_LEN(v):
        GlobalVar.VIDEO_T_LEN = v

    @staticmethod
    def get_Istrain():
        return getattr(GlobalVar, "Istrain", None)

    @staticmethod
    def set_Istrain(v):
        GlobalVar.Istrain = v

    @staticmethod
    def get_v_random_name():
        if not hasattr(GlobalVar, "encode_video_random_name"):
            GlobalVar.encode_video_random_name = "".join(random.sample("zyxwvutsrqponmlkjihgfedcba", 5))
        return GlobalVar.encode_video_random_name


_ref = sys.modules.get("global_var")
GlobalVar = _ref.GlobalVar if _ref is not None and hasattr(_ref, "GlobalVar") else _GlobalVar
