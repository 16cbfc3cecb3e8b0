"""CPU oracle for the UNITE stage-1/2/3 training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``unite_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the timed CPU baseline.
"""
