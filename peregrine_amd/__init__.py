"""peregrine_amd -- MI355X-native SHIMMER index + read-overlap hot path."""
__version__ = "0.1.0"
