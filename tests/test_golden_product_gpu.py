"""SURVEY 8(f1)-(f4) golden checks of the PRODUCT on the GPU box.  The loader (g17), native tracker (g13), evaluator (g15) and augmentation
state (g14) parity tests are host-side code and live in the CPU-marked files; the fixtures travel with the repo (the reference does not),
so the same functions are collected here a second time under the ``gpu`` marker -- the driver's ``pytest -m gpu`` run on the MI355X box
then covers the f-rows against the reference-recorded vectors as well, with ``libleod_hip.so`` (tracker.cpp / coco_eval.cpp) loaded there."""
import pytest

pytestmark = pytest.mark.gpu

import test_evaluator_cpu as _te  # noqa: E402
import test_host_cpu as _th  # noqa: E402
import test_loader_cpu as _tl  # noqa: E402
from test_loader_cpu import trees  # noqa: E402,F401  (module-scoped fixture: the synthetic dataset trees)

# f1: on-disk formats + loaders against the reference's own sequence classes (g17)
test_f1_sequence_samples_match_reference_golden = _tl.test_sequence_samples_match_reference_golden
test_f1_worker_dealing_matches_reference_golden = _tl.test_worker_dealing_matches_reference_golden
test_f1_pseudo_label_dataset_round_trip = _tl.test_pseudo_label_dataset_round_trip
test_f1_read_helpers_of_a_recording = _tl.test_read_helpers_of_a_recording
test_f1_h5_frames_agree_with_raw_frames = _tl.test_h5_frames_agree_with_raw_frames_through_a_stand_in_h5py
test_f1_h5lite_reads_the_reference_container = _tl.test_h5lite_reads_the_reference_container
test_f1_h5lite_other_dtypes_missing_chunks_and_refusals = _tl.test_h5lite_other_dtypes_missing_chunks_and_refusals
# f2: tracker post-filter in C++ against the reference-recorded tracks (g13)
test_f2_native_tracker_matches_oracle = _th.test_native_tracker_matches_oracle
test_f2_event_seq_data_track_filter_matches_reference = _th.test_event_seq_data_track_filter_matches_reference
# f3: Prophesee / COCO evaluator (g15: the reference's filters, matching, records; the native AP tables against the restated COCOeval)
test_f3_product_filter_and_windows = _te.test_product_filter_and_windows
test_f3_to_prophesee_matches_reference = _te.test_to_prophesee_matches_reference
test_f3_native_tables_equal_oracle = _te.test_native_tables_equal_oracle
test_f3_evaluator_buffer_equals_oracle = _te.test_evaluator_buffer_equals_oracle
# f4: augmentation state machine and label transforms against the reference (g14; the u8 kernels are in test_kernels_gpu.py)
test_f4_augmentor_state_and_labels_match_reference = _th.test_augmentor_state_and_labels_match_reference
