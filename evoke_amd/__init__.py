"""evoke_amd -- MI355X-native engine for EVOKE's data-parallel hot path."""
__version__ = '0.1.0'
