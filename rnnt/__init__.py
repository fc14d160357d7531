"""Drop-in import shim: ``rnnt.models`` / ``rnnt.stream`` / ``rnnt.features`` / ``rnnt.transforms`` / ``rnnt.dataset``
/ ``rnnt.tokenizer`` resolve to the MI355X engine when this repository precedes the reference
checkout on ``sys.path``."""
