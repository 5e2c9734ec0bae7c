"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot path (dmMaze/comic-text-detector,
`inference.py:TextDetector.__call__`).  Nothing in the product package
(`comic-text-detector_amd/`) may import from here; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do, and only as
the checker.

Pinning status (see DESIGN.md "Oracle"):
  * network half (`net_ref.py`): PINNED -- checked bit-for-bit (fp32, CPU)
    against the reference's own torch modules imported from /root/reference
    (`ref_import.py`), and against golden vectors generated from them
    (`tests/golden/*.npz`, generator `gen_golden.py`).
  * post-processing half (`postproc_ref.py`, `cv_ref.py`): the restatement of the reference's
    control flow is PINNED against the reference's own code run with functional stand-ins for
    the missing wheels (`ref_post_import.py`, `gen_golden_post.py`, tests/golden/post_seed*.npz);
    the third-party primitives themselves stay PARITY UNPINNED at the OpenCV / pyclipper /
    shapely / torchvision boundary -- those wheels are not installed here, the reference ships
    no tests or golden vectors for them, so they follow the published algorithms and are
    cross-checked against scipy.ndimage / brute force only.
  * annotation formats (`annot_ref.py`): PINNED against the reference's own `TextBlock.to_dict`,
    `xyxy2yolo`, `get_yololabel_strings` (`gen_golden_annot.py`, tests/golden/annot_seed*.json).
"""
