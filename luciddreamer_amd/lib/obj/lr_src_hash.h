#define LR_SRC_HASH "05e8f1ce1795"
