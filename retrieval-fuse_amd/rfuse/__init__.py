"""rfuse -- runtime of the MI355X-native RetrievalFuse refinement path: ctypes binding of librfuse_hip.so (``_lib``),
tensor-level op wrappers (``ops``), the online retrieve -> attend -> refine engine (``engine``), the sharded patch
database (``database``), the BASELINE configs (``configs``) and the seeded synthetic data (``synthetic``)."""
