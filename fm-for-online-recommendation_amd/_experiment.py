"""The Criteo experiment driver shared by main_experiment.py and main_experiment_2.py.

Same flow as the reference drivers (reference main_experiment.py:23-165): read the Criteo CSVs -> ten batches of
2,500 samples -> build the five models -> pre-train each on batch 5 for 1000 mini-batch steps
(update_embedding + predict) -> the per-sample online loop run_experiment over every batch -> pickle the result
dict and the models.  The constants below are the reference's; the command line only adds what the reference lacks
(the Criteo files are not distributed with it): --synthetic writes Criteo-shaped CSVs in the reference's format
first, and the sizes can be scaled down for a quick run.
"""
import argparse
import os
import pickle
import sys
import tempfile
from time import time

import numpy as np

sys.path.append("../")

from utils import data_preprocess                                   # noqa: E402
from models.models_online_deep.deepfm_adam import DeepFMAdam         # noqa: E402
from models.models_online_deep.deepfm_onn import DeepFMOnn           # noqa: E402
from models.models_online_deep.nfm_adam import NFMAdam               # noqa: E402
from models.models_online_deep.nfm_onn import NFMOnn                 # noqa: E402
from models.models_online_deep.fm_adam import FMAdam                 # noqa: E402

TRAIN_CSV = "dataset/criteo/tiny_train_input.csv"
EMB_CSV = "dataset/criteo/category_emb.csv"

num_hidden_layers = 5
neuron_per_hidden_layer = 10
data_feature_dim = 39
embedding_size = 10
n = 0.0001

feature_sizes = [63, 113, 126, 51, 224, 148, 100, 79, 104, 9, 32, 57, 82, 1457, 555, 176373, 129683, 305, 19, 11887,
                 632, 3, 41738, 5170, 175446, 3170, 27, 11356, 165602, 10, 4641, 2030, 4, 172761, 18, 15, 57903, 86,
                 44549]


def write_synthetic_criteo(directory, n_samples, seed=0):
    """Criteo-shaped files in the reference's on-disk format: `field,category,index` and `label,i0..i38`."""
    rng = np.random.default_rng(seed)
    emb = os.path.join(directory, "category_emb.csv")
    csv = os.path.join(directory, "tiny_train_input.csv")
    with open(emb, "w") as fh:
        for f, size in enumerate(feature_sizes):
            fh.write("".join(f"{f},c{c},{c}\n" for c in range(size)))
    cols = np.stack([rng.integers(0, s, size=n_samples) for s in feature_sizes], axis=1)
    labels = (rng.uniform(size=n_samples) < 0.5).astype(int)
    with open(csv, "w") as fh:
        for lab, row in zip(labels, cols):
            fh.write(str(lab) + "," + ",".join(map(str, row)) + "\n")
    return csv, emb


def build_models():
    deep = dict(embedding_size=embedding_size, num_hidden_layers=num_hidden_layers,
                neuron_per_hidden_layer=neuron_per_hidden_layer, n=n)
    return [DeepFMAdam(feature_sizes, **deep), DeepFMOnn(feature_sizes, **deep), NFMAdam(feature_sizes, **deep),
            NFMOnn(feature_sizes, **deep), FMAdam(feature_sizes, embedding_size=embedding_size, n=n)]


def run(data_config, model_pickle_by_str=False, argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", action="store_true", help="generate Criteo-shaped CSVs instead of reading dataset/criteo")
    ap.add_argument("--num-batchdata", type=int, default=2500)
    ap.add_argument("--num-batch", type=int, default=10)
    ap.add_argument("--pretrain-iters", type=int, default=1000)
    ap.add_argument("--out", default=os.getcwd() + "/performance/")
    args = ap.parse_args(argv)
    num_batchdata, num_batch = args.num_batchdata, args.num_batch
    save_log, save_model = args.out + "save_log/", args.out + "save_model/"
    os.makedirs(save_log, exist_ok=True)
    os.makedirs(save_model, exist_ok=True)

    train_csv, emb_csv = TRAIN_CSV, EMB_CSV
    if args.synthetic:
        tmp = tempfile.mkdtemp(prefix="fmx_criteo_")
        train_csv, emb_csv = write_synthetic_criteo(tmp, 4 * num_batch * num_batchdata)
    train_dict = data_preprocess.read_criteo_data(train_csv, emb_csv)
    print("train size :", train_dict["size"])

    if data_config == "Iteration":
        batches = data_preprocess.create_ten_iter(train_csv, emb_csv, num_batch, num_batchdata)
    elif isinstance(data_config, int):
        ratio = data_config if model_pickle_by_str else int(num_batch / data_config)
        batches = data_preprocess.create_dataset(train_csv, emb_csv, ratio, num_batch, num_batchdata)
    else:
        batches = data_preprocess.create_dataset(train_csv, emb_csv, int(num_batch / 2), num_batch, num_batchdata)
    batch_train_Xi_list, batch_train_Xv_list, batch_train_Y_list, ratio_list = batches

    model_list = build_models()
    model_name_list = [str(model).split("-")[0] for model in model_list]
    print(model_name_list)

    # ---- pre-training (reference :92-105) ----
    mid = int(num_batch / 2)
    for ith_model, ith_model_name in zip(model_list, model_name_list):
        print(f"====={ith_model_name}=====")
        for j in range(args.pretrain_iters):
            loss_emb = ith_model.update_embedding(batch_train_Xi_list[mid], batch_train_Xv_list[mid], batch_train_Y_list[mid])
            pred_label = ith_model.predict(batch_train_Xi_list[mid], batch_train_Xv_list[mid])
            if j % 100 == 0:
                print("i th iter %d , loss : %f" % (j, loss_emb.cpu().data))
                right_count = len((np.where(np.asarray(pred_label) == np.asarray(batch_train_Y_list[mid])))[0])
                total_count = len(np.asarray(batch_train_Y_list[mid]))
                print("training accuracy : %.4f\n" % (right_count / total_count))

    # ---- the online experiment (reference :111-145) ----
    result_dict = {"roc": {}, "data_ratio": {}, "time": {}, "accuracy": {}, "num_batch": num_batch,
                   "num_batchdata": num_batchdata, "user_auc_mean": {}}
    for ith_exp in range(num_batch):
        print("#" * 100)
        for jth_model_name, jth_model in zip(model_name_list, model_list):
            print("%d th batch, %s model" % (ith_exp + 1, jth_model_name))
            print("neg ratio : %d,  pos ratio %d " % (ratio_list[ith_exp][0], ratio_list[ith_exp][1]))
            time_elapsed, accuracy, roc, confusion_matrix = jth_model.run_experiment(
                batch_train_Xi_list[ith_exp], batch_train_Xv_list[ith_exp], batch_train_Y_list[ith_exp])
            print("fpr : %.4f , tpr : %.4f " % (roc["fpr"], roc["tpr"]))
            print("confusion matrix : %s" % confusion_matrix)
            print("accuracy : %.4f \n" % accuracy)
            for key, val in (("roc", roc), ("data_ratio", ratio_list[ith_exp]), ("time", time_elapsed),
                             ("accuracy", accuracy)):
                result_dict[key].setdefault(jth_model_name, []).append(val)

    stamp = str(int(time())) if model_pickle_by_str else str(time())
    save_filename = ("Time_Stamp" + stamp + "-Dataset" + str("criteo") + "-Num_BatchLength" + str(num_batchdata)
                     + "-Num_Batch" + str(num_batch) + "-Num_Hidden_Layers" + str(num_hidden_layers)
                     + "-Neuron_Per_Hidden_Layer" + str(neuron_per_hidden_layer) + "_" + str(data_config))
    with open(save_log + save_filename + ".pickle", "wb") as f:
        pickle.dump(result_dict, f)
    for ith_model, ith_model_name in zip(model_list, model_name_list):
        name = ith_model_name + "_" + str(data_config)
        with open(save_model + name + ".pickle", "wb") as f:
            pickle.dump(ith_model, f)
    print("save_log : %s" % (save_log))
    print("save_model : %s" % (save_model))
    return result_dict
