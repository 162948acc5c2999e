"""Gateway-API data plane: Envoy external-processing (ext_proc) endpoint picker."""
