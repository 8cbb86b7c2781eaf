"""CPU oracle for the FPN Mask R-CNN training path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a float32, step-for-step NumPy restatement of the reference algorithm
(katotetsuro/chainer-maskrcnn plus the third-party semantics it calls into:
Chainer / ChainerCV / OpenCV / the un-vendored ``roi_align`` submodule).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it -- and there only as the checker, never as the thing measured or
shipped.  The product path (``chainer-maskrcnn_amd/``) never imports it and
fails loudly when the HIP library is missing.

PARITY STATUS (see DESIGN.md "Oracle"):
  * The reference has no tests, no golden vectors and none of its third-party
    dependencies can be installed here, and its ROIAlign operator lives in a git
    submodule that is absent from /root/reference.  For those pieces this oracle
    is a restatement of the published algorithm => "parity unpinned".
  * What *is* pinned by reference code executed in the build container
    (tests/golden/make_reference_vectors.py): ``map_rois_to_fpn_levels``
    (arithmetic) and the control flow of ``ProposalTargetCreator`` (sampling
    sizes, label shifting, index bookkeeping) with this oracle's box utilities
    injected for the absent ChainerCV / OpenCV calls.

Every function cites the reference file:line (relative to /root/reference) or
the third-party routine it follows.
"""
