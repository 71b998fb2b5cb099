"""Test infrastructure only: CPU oracle of the reference algorithm. Never imported by hybridgl_amd/."""
