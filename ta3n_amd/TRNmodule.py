"""Temporal relation modules - parameter containers with the reference's names
(TRNmodule.py:27-56) and the bit-exact frame-tuple tables.

Inside `VideoModel` the multi-scale relation network runs as one grouped GEMM
launch of libta3n_hip.so (the per-tuple gather+concat of TRNmodule.py:60-63,
75-77 is folded into the A-operand addressing; see csrc/ta3n_plan.cpp phase F2),
so this module's own `forward` is never called on main.py's path.
"""
from __future__ import annotations

from math import comb

import torch
from torch import nn

from . import _lib


class RelationModuleMultiScale(nn.Module):
    """Multi-scale TRN (TRNmodule.py:27-86): scales [T..2], at most 3 evenly spaced frame
    tuples per scale, one ReLU-Linear(scale*D -> bottleneck)-ReLU fusion per scale."""

    def __init__(self, img_feature_dim, num_bottleneck, num_frames, verbose=True):
        super().__init__()
        self.subsample_num = 3                                               # TRNmodule.py:32
        self.img_feature_dim = img_feature_dim
        self.num_frames = num_frames
        self.scales = [i for i in range(num_frames, 1, -1)]                  # :34
        # the reference enumerates all C(T,s) tuples (:36-41); only the selected ones are ever
        # used (:60, :71) and they are produced by unranking in the C library
        self.relations_selected = _lib.relation_table(num_frames)
        self.subsample_scales = [min(self.subsample_num, comb(num_frames, s)) for s in self.scales]
        self.fc_fusion_scales = nn.ModuleList(                               # :44-54 (default nn.Linear init)
            nn.Sequential(nn.ReLU(), nn.Linear(s * img_feature_dim, num_bottleneck), nn.ReLU()) for s in self.scales)
        if verbose:
            print('Multi-Scale Temporal Relation Network Module in use', ['%d-frame relation' % i for i in self.scales])

    def return_relationset(self, num_frames, num_frames_relation):           # :84-86 (utility)
        import itertools
        return list(itertools.combinations(range(num_frames), num_frames_relation))

    def forward(self, input):
        raise NotImplementedError(
            "RelationModuleMultiScale runs inside ta3n_amd.models.VideoModel as a grouped HIP GEMM; a standalone "
            "forward is not part of the TA3N train-step path (SURVEY.md 8)")


class RelationModule(nn.Module):
    """Single-scale TRN (TRNmodule.py:6-25).  The reference's 'trn' mode crashes in
    VideoModel.forward (models.py:639 uses relation_domain_classifier_all, built only for
    'trn-m'), so only the parameter container is provided."""

    def __init__(self, img_feature_dim, num_bottleneck, num_frames):
        super().__init__()
        self.num_frames, self.img_feature_dim, self.num_bottleneck = num_frames, img_feature_dim, num_bottleneck
        self.classifier = nn.Sequential(nn.ReLU(), nn.Linear(num_frames * img_feature_dim, num_bottleneck), nn.ReLU())

    def forward(self, input):
        raise NotImplementedError("frame_aggregation='trn' is not on the TA3N hot path")
