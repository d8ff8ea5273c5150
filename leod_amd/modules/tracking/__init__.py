"""Tracking post-filter of the pseudo-label loop (reference: modules/tracking/, 436 lines of NumPy/Python loops).

The whole per-recording tracker -- linear-velocity tracklets, confidence-ordered greedy IoU association, short-tracklet
removal, in-painting of missed detections -- is ONE call into the native library (``leod_track_filter``,
``leod_amd/csrc/tracker.cpp``); this module is the array plumbing around it.  Host code: no GPU involved.
"""
from .linear import track, track_filter, LinearTrackerConfig  # noqa: F401
