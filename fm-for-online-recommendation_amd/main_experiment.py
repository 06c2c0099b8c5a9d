"""Entry point kept from the reference (main_experiment.py): the Criteo online experiment with the "Iteration"
batch schedule (batch i holds (i+1)/10 positives).  `python main_experiment.py [--synthetic ...]`."""
from _experiment import run

data_config = "Iteration"

if __name__ == "__main__":
    run(data_config)
