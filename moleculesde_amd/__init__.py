"""moleculesde_amd — MI355X-native hot path of MoleculeSDE pretraining."""
__version__ = "0.1.0"
