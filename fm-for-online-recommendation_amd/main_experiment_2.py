"""Entry point kept from the reference (main_experiment_2.py): as main_experiment.py with a fixed 7:3 class ratio
(data_config = 7 -> create_dataset(..., batch_ratio=7, ...), reference main_experiment_2.py:33,40-42)."""
from _experiment import run

data_config = 7

if __name__ == "__main__":
    run(data_config, model_pickle_by_str=True)
