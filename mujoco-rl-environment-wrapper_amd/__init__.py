"""MI355X-native batched multi-agent env stepper behind the MuJoCoRL plugin surface.

Loaded under the import name ``mjrl_amd`` (the directory name carries hyphens; see
``__graft_entry__.load_package``).
"""
