"""TEST INFRASTRUCTURE ONLY -- CPU restatements ("oracles") of the OpenGaussian hot path.

Nothing in the shipped product (``opengaussian_amd/``) may import from this package.
Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- and there only as the checker / the reported CPU baseline,
never as the thing measured or shipped.
"""
