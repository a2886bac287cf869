"""CPU oracle for the ColPali late-interaction path -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
the product package (morphik-core_amd/).
"""
