"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see cpu_ref.py header)."""
