"""cpflow_amd — MI355X-native engine for cpflow's CP-template synthesis hot path."""
__version__ = '0.1.0'
