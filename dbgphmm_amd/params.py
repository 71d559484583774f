"""PHMMParams: host-side mirror of the reference's parameter struct.

Follows /root/reference/src/hmmv2/params.rs:16-125.  Every ``p_*`` field is an f64
LOG probability (the reference's ``Prob``); integer knobs keep the reference names.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass, replace

MAX_ACTIVE_NODES = 400  # src/hmmv2/table.rs:22
MAX_DEL = 4  # src/hmmv2/table.rs:17


def _ln(x: float) -> float:
    return math.log(x) if x > 0.0 else -math.inf


class CPHMMParams(ctypes.Structure):
    """C layout shared by include/phmm_amd.h (phmm_params) and oracle/phmm_oracle.h."""

    _fields_ = [(n, ctypes.c_double) for n in (
        "p_mismatch", "p_match", "p_random", "p_gap_open", "p_gap_ext", "p_end",
        "p_MM", "p_IM", "p_DM", "p_MI", "p_II", "p_DI", "p_MD", "p_ID", "p_DD")] + [
        ("n_active_nodes", ctypes.c_int64),
        ("active_node_max_ratio", ctypes.c_double),
        ("n_warmup", ctypes.c_int64),
        ("warmup_threshold", ctypes.c_int64),
        ("n_max_gaps", ctypes.c_int64),
    ]


@dataclass(frozen=True)
class PHMMParams:
    p_mismatch: float
    p_match: float
    p_random: float
    p_gap_open: float
    p_gap_ext: float
    p_end: float
    p_MM: float
    p_IM: float
    p_DM: float
    p_MI: float
    p_II: float
    p_DI: float
    p_MD: float
    p_ID: float
    p_DD: float
    n_active_nodes: int = 40
    active_node_max_ratio: float = 30.0
    n_warmup: int = 50
    warmup_threshold: int = MAX_ACTIVE_NODES // 2
    n_max_gaps: int = 4

    @staticmethod
    def new(p_mismatch: float, p_gap_open: float, p_gap_ext: float, p_end: float,
            n_active_nodes: int, n_warmup: int) -> "PHMMParams":
        """params.rs:73-113; arguments are linear probabilities."""
        assert n_active_nodes > 0
        assert n_warmup > 0
        assert n_active_nodes < MAX_ACTIVE_NODES
        l_mis, l_go, l_ge, l_end = _ln(p_mismatch), _ln(p_gap_open), _ln(p_gap_ext), _ln(p_end)
        # the reference round-trips through Prob::to_value() = exp(ln p)
        go, ge, pe, mis = (math.exp(v) if v > -math.inf else 0.0 for v in (l_go, l_ge, l_end, l_mis))
        return PHMMParams(
            p_mismatch=l_mis, p_gap_open=l_go, p_gap_ext=l_ge, p_end=l_end,
            p_DD=l_ge, p_II=l_ge, p_MI=l_go, p_MD=l_go, p_ID=l_go, p_DI=l_go,
            p_MM=_ln(1.0 - 2.0 * go - pe),
            p_DM=_ln(1.0 - go - ge - pe),
            p_IM=_ln(1.0 - go - ge - pe),
            p_match=_ln(1.0 - mis),
            p_random=_ln(0.25),
            n_active_nodes=n_active_nodes, active_node_max_ratio=30.0, n_warmup=n_warmup,
            n_max_gaps=4, warmup_threshold=MAX_ACTIVE_NODES // 2)

    @staticmethod
    def uniform(p: float) -> "PHMMParams":
        """params.rs:116-125"""
        return PHMMParams.new(p, p, p, 0.00001, 40, 50)

    @staticmethod
    def default() -> "PHMMParams":
        return PHMMParams.uniform(0.01)

    @staticmethod
    def mid_error_2() -> "PHMMParams":
        return PHMMParams.uniform(0.02)

    @staticmethod
    def mid_error() -> "PHMMParams":
        return PHMMParams.uniform(0.05)

    @staticmethod
    def high_error() -> "PHMMParams":
        return PHMMParams.uniform(0.1)

    @staticmethod
    def zero_error() -> "PHMMParams":
        return PHMMParams.uniform(0.0)

    def with_(self, **kw) -> "PHMMParams":
        return replace(self, **kw)

    def to_c(self) -> CPHMMParams:
        c = CPHMMParams()
        for name, _ in CPHMMParams._fields_:
            setattr(c, name, getattr(self, name))
        return c
