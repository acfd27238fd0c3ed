"""CPU oracle for the CMGAN generator forward path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package (``cmgan_amd``) may
import from here; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker / the
timed CPU baseline - never as the product path.
"""
