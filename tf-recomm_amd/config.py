"""Hyper-parameters of the training driver (reference: config.py:4-17, plus the per-dataset
keys USER_NUM / ITEM_NUM / BATCH_SIZE the fork reads from data/<name>/config.yml).

Defaults are the canonical MovieLens-1M SVD of README.md:31-57 at BASELINE.json config 1.
"""
DIM = 15
EPOCH_MAX = 100
BATCH_SIZE = 1000
LEARNING_RATE = 1e-3          # Adam (README.md:39); the fork's SGD value is 5e-3 (config.py:8)
LAMBDA_REG = 0.05
DISCRETE = False              # True -> the fork's binary-outcome variant (nll, |item|, bias l2, SGD)
DEVICE = 0                    # HIP device ordinal (the reference's "/cpu:0" | "/gpu:0" string)
USER_NUM = 6040
ITEM_NUM = 3952
SEED = 13575                  # svd_train_val.py:15
