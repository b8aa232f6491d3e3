"""groma_amd: MI355X-native implementation of Groma's localized-visual-tokenization forward path."""
__version__ = "0.1.0"
