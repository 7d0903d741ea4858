"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the BayesSim posterior-estimator training path of
NVlabs/bayes-sim-ig (summarizers -> RFF -> MDN forward / NLL / fit).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package, and only as the checker /
reported baseline.  Nothing under ``bayes_sim_ig_amd/`` imports it; the
product path fails loudly when the HIP library is missing.
"""
