"""Drop-in import path: with this repository's root on sys.path in place of the reference's, the reference's
callers (`apps/train_gcn.py`, `apps/eval_interhand.py`, `core/gcn_trainer.py`) import `models.model`,
`models.manolayer`, `models.encoder`, `models.decoder` from here and get the MI355X-native implementations
in `renderih_amd/`."""
