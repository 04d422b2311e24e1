"""CPU oracle for the PosteriFlow hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / the reported CPU
baseline.  Nothing under ``posteriflow_amd/`` imports it; the product path
raises if the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):

* ``oracle.nflows_restated`` restates the third-party library ``nflows``
  (un-vendored, un-pinned: ``/root/reference/environment.yaml:35``; latest
  public release 0.14).  nflows is absent from this image and the reference
  ships no test, golden vector or checkpoint for the flow, so for the
  MADE / rational-quadratic-spline transform **parity is unpinned**; it is
  anchored instead on the reference's call sites (``src/ahsd/models/flows.py``)
  and on the known-answer tests in ``tests/test_oracle_kat.py``.
* ``oracle.flow_ref`` / ``oracle.lean_ref`` restate the reference's own code
  (``flows.py``, ``lean_npe.py``, ``coherent_encoder.py``) and ARE pinned by the
  golden vectors in ``tests/golden/`` that ``tests/golden/make_golden.py``
  produced by running the reference's classes in the build container.
"""
