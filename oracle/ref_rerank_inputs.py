"""Literal restatement of the LLM re-ranker's input construction -- TEST INFRASTRUCTURE ONLY.

Follows ``get_inputs`` in ``experiments/profile/stress_test.py:97-134`` (the in-repo restatement of FlagLLMReranker's
preprocessing): it keeps the reference's calls into the tokenizer object -- ``tokenizer(...)``,
``tokenizer.prepare_for_model(..., truncation='only_second')``, ``tokenizer.pad(..., pad_to_multiple_of=8)`` -- so the
product's hand-built version (veritasfi_amd.encoder.build_llm_reranker_inputs, which only needs ``__call__`` and
``bos_token_id``) can be compared with it on any tokenizer implementing those methods.
"""

DEFAULT_PROMPT = ("Given a query A and a passage B, determine whether the passage contains an answer to the query by "
                  "providing a prediction of either 'Yes' or 'No'.")


def get_inputs(pairs, tokenizer, prompt=None, max_length=1024):
    if prompt is None:
        prompt = DEFAULT_PROMPT
    sep = "\n"
    prompt_inputs = tokenizer(prompt, return_tensors=None, add_special_tokens=False)["input_ids"]      # :102-104
    sep_inputs = tokenizer(sep, return_tensors=None, add_special_tokens=False)["input_ids"]            # :105-107
    inputs = []
    for query, passage in pairs:
        query_inputs = tokenizer(f"A: {query}", return_tensors=None, add_special_tokens=False,
                                 max_length=max_length * 3 // 4, truncation=True)                      # :110-114
        passage_inputs = tokenizer(f"B: {passage}", return_tensors=None, add_special_tokens=False,
                                   max_length=max_length, truncation=True)                             # :115-119
        item = tokenizer.prepare_for_model([tokenizer.bos_token_id] + query_inputs["input_ids"],
                                           sep_inputs + passage_inputs["input_ids"], truncation="only_second",
                                           max_length=max_length, padding=False, return_attention_mask=False,
                                           return_token_type_ids=False, add_special_tokens=False)     # :120-129
        item["input_ids"] = item["input_ids"] + sep_inputs + prompt_inputs                             # :130
        item["attention_mask"] = [1] * len(item["input_ids"])                                          # :131
        inputs.append(item)
    return tokenizer.pad(inputs, padding=True, max_length=max_length + len(sep_inputs) + len(prompt_inputs),
                         pad_to_multiple_of=8, return_tensors="np")                                    # :134-140
